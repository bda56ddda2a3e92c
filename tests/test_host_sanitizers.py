"""The host-side file readers under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build; the pool has no GPU sanitizers).

tests/sanitize/host_fuzz.cpp links the product's own host sources -- the pbrt / pbf / ply readers, scene conversion, the BVH builder
and every image decoder (.hdr .pfm .png .tga .jpg .bmp .dds) -- and reads each fixture as it is and then a few hundred damaged copies
(bit flips, truncations, inserted runs, huge length fields, zeroed ranges; for scene text also token-level edits).  A reader may
succeed or refuse; the sanitizer runtime aborts on any out-of-bounds access, use after free, leak, signed overflow or bad shift.
Findings of the first runs, all fixed: a JPEG plane copy that trusted inexact sampling ratios, unchecked frame-header lengths,
allocations sized by a header before the payload was known to exist (DDS, TGA, PNG, HDR, PFM, PLY), fopen() of a directory, mesh
indices beyond the vertex array, PLY normals without an `nx` property, an object instanced inside its own definition."""
import glob
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
BUILD = os.path.join(HERE, "sanitize", "_build")


@pytest.fixture(scope="module")
def host_fuzz():
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no g++ / make")
    r = subprocess.run(["make", "-C", os.path.join(HERE, "sanitize"), "OUT=" + BUILD, "-j4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return os.path.join(BUILD, "host_fuzz")


def _run(exe, seed, mutations, files):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, str(seed), str(mutations)] + files, capture_output=True, text=True, timeout=900, env=env)
    tail = (r.stdout + r.stderr)[-6000:]
    assert r.returncode == 0 and r.stdout.rstrip().endswith("ok"), tail
    lines = [l for l in r.stdout.splitlines() if ": " in l]
    assert len(lines) == len(files) and all("reads as it is" in l for l in lines), tail   # every undamaged fixture decodes
    return lines


def test_image_readers_survive_damaged_files(host_fuzz):
    files = sorted(glob.glob(os.path.join(GOLDEN, "images_r3", "*.jpg")) + glob.glob(os.path.join(GOLDEN, "images_r3", "*.bmp")) + glob.glob(os.path.join(GOLDEN, "images_r3", "*.dds")) +
                   glob.glob(os.path.join(GOLDEN, "images", "*.png")) + glob.glob(os.path.join(GOLDEN, "images", "*.tga")) + [os.path.join(GOLDEN, "scenes", "Teapot", "textures", "sky.hdr")])
    assert len(files) > 60
    _run(host_fuzz, 7, 150, files)


def test_scene_readers_survive_damaged_files(host_fuzz, tmp_path):
    scenes = str(tmp_path / "scenes"); shutil.copytree(os.path.join(GOLDEN, "scenes"), scenes)      # damaged copies are written beside the originals
    shutil.copy(os.path.join(GOLDEN, "cornell-box.pbf"), scenes)
    rng = np.random.default_rng(3)
    v = rng.random((12, 8)).astype("<f4"); f = rng.integers(0, 12, (9, 3)).astype("<i4")
    hdr = ("ply\nformat binary_little_endian 1.0\nelement vertex 12\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\n"
           "property float u\nproperty float v\nelement face 9\nproperty list uint8 int vertex_indices\nend_header\n")
    open(os.path.join(scenes, "small_bin.ply"), "wb").write(hdr.encode() + v.tobytes() + b"".join(b"\x03" + r.tobytes() for r in f))
    open(os.path.join(scenes, "small_ascii.ply"), "w").write(hdr.replace("binary_little_endian", "ascii") + "\n".join(" ".join("%g" % x for x in r) for r in v) + "\n" + "\n".join("3 %d %d %d" % tuple(r) for r in f) + "\n")
    files = [os.path.join(scenes, p) for p in ("small_bin.ply", "small_ascii.ply", "cornell-box/scene.pbrt", "alpha-card/instanced.pbrt", "alpha-card/scene.pbrt", "instances/scene.pbrt",
                                                "material-maps/scene.pbrt", "mix-glass/scene.pbrt", "furnace/slab.pbrt", "cornell-box.pbf")]
    _run(host_fuzz, 5, 200, files)


def test_hostile_scene_constructs_are_refused(built, tmp_path):
    """the product library itself (no sanitizer): the constructs the fuzzing found are refused with a message, not executed"""
    from tracerboy_amd import api
    cam = 'LookAt 0 0 5 0 0 0 0 1 0\nCamera "perspective" "float fov" [40]\nWorldBegin\n'
    tri = 'Shape "trianglemesh" "point P" [0 0 0 1 0 0 0 1 0] "integer indices" [0 1 2]\n'
    cases = {
        "self_instance.pbrt": cam + 'ObjectBegin "a"\n' + tri + 'ObjectInstance "a"\nObjectEnd\nObjectInstance "a"\nWorldEnd\n',
        "index_out_of_range.pbrt": cam + 'Shape "trianglemesh" "point P" [0 0 0 1 0 0 0 1 0] "integer indices" [0 1 7]\nWorldEnd\n',
        "few_normals.pbrt": cam + 'Shape "trianglemesh" "point P" [0 0 0 1 0 0 0 1 0] "normal N" [0 0 1] "integer indices" [0 1 2]\nWorldEnd\n',
    }
    for name, text in cases.items():
        q = str(tmp_path / name); open(q, "w").write(text)
        with pytest.raises(api.TracerBoyError):
            api.HostScene(q)
    # nested instancing multiplies: eight levels of eight instances each are 8^8 = 16.7 M shapes, beyond 2^24 - 1
    text = cam + 'ObjectBegin "o0"\n' + tri + "ObjectEnd\n"
    for k in range(1, 9):
        text += 'ObjectBegin "o%d"\n' % k + ('ObjectInstance "o%d"\n' % (k - 1)) * 8 + "ObjectEnd\n"
    text += 'ObjectInstance "o8"\nWorldEnd\n'
    q = str(tmp_path / "multiplying.pbrt"); open(q, "w").write(text)
    for flatten in (True, False):
        with pytest.raises(api.TracerBoyError):
            api.HostScene(q, flatten_instances=flatten)
    # an image whose header promises more than the file holds is refused before anything of that size is allocated
    q = str(tmp_path / "huge.tga"); open(q, "wb").write(struct.pack("<BBBHHBHHHHBB", 0, 0, 2, 0, 0, 0, 0, 0, 16000, 16000, 32, 8) + b"\0" * 64)
    with pytest.raises(api.TracerBoyError):
        api.DecodeImage(q)
    os.mkdir(str(tmp_path / "dir.png"))
    with pytest.raises(api.TracerBoyError):
        api.DecodeImage(str(tmp_path / "dir.png"))                   # a directory is not an image
