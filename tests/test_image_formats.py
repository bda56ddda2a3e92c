"""JPEG / BMP / DDS texture decoders (tracerboy_amd/csrc/host/image_formats.cpp; the WIC and DDS branches of
TracerBoy.cpp:2218-2226) against the fixtures of tests/golden/make_image_fixtures_r3.py.

JPEG: sequential AND progressive files written by Pillow -- 4:4:4, 4:2:2, 4:2:0, grey, optimised Huffman tables, restart intervals, odd
sizes, a 1 x 1 image -- against Pillow's (libjpeg-turbo's) own decode: the decoder restates the IJG arithmetic (slow-integer IDCT, fancy
upsampling, fixed-point colour conversion), so the bar is EQUALITY of the 8-bit samples.
BMP: 1 / 4 / 8-bit palettes, 16-bit 5-6-5 bit fields (top-down), 24-bit, 32-bit with and without an alpha mask.
DDS: BC1-BC5 (incl. signed BC4 / BC5 and a DX10 _SRGB header), 32 / 24 / 16-bit masks (16-bit formats expanded like
DDS_FLAGS_NO_16BPP), L8 / A8L8, RGBA16F, RGBA32F, RGBA8 through the DX10 header."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
IMAGES = os.path.join(GOLDEN, "images_r3")
EXPECTED = np.load(os.path.join(IMAGES, "expected_r3.npz"))


@pytest.mark.parametrize("name", sorted(EXPECTED.files))
def test_decoder_matches_fixture(built, name):
    from tracerboy_amd import api
    img, normalized, has_alpha = api.DecodeImage(os.path.join(IMAGES, name))
    want = EXPECTED[name]
    assert img.shape == want.shape
    assert normalized == (name not in ("rgba16f.dds", "rgba32f_dx10.dds"))
    if name.endswith(".dds") and name.startswith("bc"):
        # interpolated palette entries: the same float32 expressions on both sides, up to the order of two roundings
        assert np.allclose(img, want, rtol=0, atol=2e-7), float(np.abs(img - want).max())
    else:
        assert np.array_equal(img.view(np.uint32), want.view(np.uint32)), (name, float(np.abs(img - want).max()) * 255)
    assert has_alpha == bool(np.any(want[..., 3] != 1.0))


def test_damaged_files_are_refused(built, tmp_path):
    from tracerboy_amd import api
    src = open(os.path.join(IMAGES, "q75_420.jpg"), "rb").read()
    for cut in (len(src) // 3, len(src) - 40):                        # truncated entropy-coded data: decodes garbage or throws, never crashes;
        q = str(tmp_path / ("cut%d.jpg" % cut)); open(q, "wb").write(src[:cut])
        try: api.DecodeImage(q)
        except api.TracerBoyError: pass
    for name in ("bc3.dds", "rgb24.bmp", "a8r8g8b8.dds"):
        src = open(os.path.join(IMAGES, name), "rb").read()
        q = str(tmp_path / ("short_" + name)); open(q, "wb").write(src[:len(src) // 2])
        with pytest.raises(api.TracerBoyError):
            api.DecodeImage(q)


def test_scene_with_jpeg_texture_loads(built, tmp_path):
    """a pbrt scene whose material uses a .jpg imagemap (vw-van-style content): the texel pool holds the decoded texels"""
    import ctypes as C
    import shutil
    from tracerboy_amd import api
    shutil.copy(os.path.join(IMAGES, "q85_422.jpg"), tmp_path / "albedo.jpg")
    scene = '''
LookAt 0 1 5  0 1 0  0 1 0
Camera "perspective" "float fov" [40]
Film "image" "integer xresolution" [64] "integer yresolution" [48]
WorldBegin
Texture "tex-a" "spectrum" "imagemap" "string filename" ["albedo.jpg"]
MakeNamedMaterial "A" "string type" ["matte"] "texture Kd" ["tex-a"]
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [10 10 10]
  Shape "trianglemesh" "integer indices" [0 1 2] "point P" [-1 3 -1  1 3 -1  0 3 1] "float uv" [0 0 1 0 0 1]
AttributeEnd
NamedMaterial "A"
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-2 0 -2  2 0 -2  2 0 2  -2 0 2] "float uv" [0 0 1 0 1 1 0 1]
WorldEnd
'''
    p = tmp_path / "scene.pbrt"; p.write_text(scene)
    hs = api.HostScene(str(p))
    v = hs.view()
    assert v.numImages == 1
    a, _, has_alpha = api.DecodeImage(str(tmp_path / "albedo.jpg"))
    d = v.images[0]
    pool = np.ctypeslib.as_array(C.cast(v.texelPool, C.POINTER(C.c_float)), shape=((d.texelOffset + d.width * d.height) * 4,))
    assert (d.width, d.height) == (67, 45) and np.array_equal(pool[d.texelOffset * 4:].reshape(-1, 4), a.reshape(-1, 4)) and not has_alpha


@pytest.mark.parametrize("name", ["bc7_modes.dds", "bc7_odd.dds"])
def test_bc7_blocks_match_an_independent_decoder(built, name):
    """1024 random BC7 blocks, 120 per mode (tests/golden/make_bc7_fixture.py), against Pillow's decode of the same file: BC7 is
    integer arithmetic end to end, so the 8-bit values must be EQUAL.  The reserved encoding (low byte 0) decodes to (0,0,0,0) as the
    format specifies and DirectXTex does; Pillow leaves alpha at 255 there -- those blocks are compared on RGB and checked for alpha 0."""
    from tracerboy_amd import api
    want = np.load(os.path.join(IMAGES, "expected_bc7.npz"))[name]
    img, normalized, has_alpha = api.DecodeImage(os.path.join(IMAGES, name))
    assert normalized and img.shape == want.shape
    raw = open(os.path.join(IMAGES, name), "rb").read()[148:]
    h, w = want.shape[:2]; bw = (w + 3) // 4
    reserved = np.zeros((h, w), bool)
    for by in range((h + 3) // 4):
        for bx in range(bw):
            if raw[(by * bw + bx) * 16] == 0: reserved[by * 4:by * 4 + 4, bx * 4:bx * 4 + 4] = True
    assert reserved.sum() == (64 * 16 if name == "bc7_modes.dds" else 0)
    expect = want.astype(np.float32) / np.float32(255)
    expect[reserved] = 0.0
    assert np.array_equal(img.view(np.uint32), expect.view(np.uint32))
    assert has_alpha
