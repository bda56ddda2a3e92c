"""PNG / TGA texture decoders (tracerboy_amd/csrc/host/image_decode.cpp) against the fixtures written by
tests/golden/make_image_fixtures.py: every colour type / bit depth / filter type / interlace mode of PNG, stored, fixed and
dynamic deflate blocks, TGA true-colour / grey, raw and run-length, both row orders.  Expected texels are the DXGI typed
load of what DirectXTex would produce (grey -> (g, 0, 0, 1), UNORM scaling, alpha 1 where absent)."""
import os
import zlib

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
IMAGES = os.path.join(GOLDEN, "images")
EXPECTED = np.load(os.path.join(IMAGES, "expected.npz"))


@pytest.mark.parametrize("name", sorted(EXPECTED.files))
def test_decoder_matches_fixture(built, name):
    from tracerboy_amd import api
    img, normalized, has_alpha = api.DecodeImage(os.path.join(IMAGES, name))
    want = EXPECTED[name]
    assert img.shape == want.shape and normalized
    assert np.array_equal(img.view(np.uint32), want.view(np.uint32))
    assert has_alpha == bool(np.any(want[..., 3] != 1.0))


def test_decoder_reads_what_the_writer_writes(built, tmp_path):
    """the output stage's PNG writer (stored deflate blocks) round-trips through the texture decoder"""
    from tracerboy_amd import api
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (130, 129, 4), dtype=np.uint8)
    p = str(tmp_path / "w.png")
    api.WriteImage(p, img)
    got, normalized, has_alpha = api.DecodeImage(p)
    assert np.array_equal(got, img.astype(np.float32) / np.float32(255)) and has_alpha and normalized


def test_decoder_rejects_damaged_files(built, tmp_path):
    from tracerboy_amd import api
    src = open(os.path.join(IMAGES, "rgba8.png"), "rb").read()
    bad = bytearray(src); bad[len(bad) // 2] ^= 0x55          # corrupt the compressed stream / its checksum
    p = str(tmp_path / "bad.png"); open(p, "wb").write(bytes(bad))
    with pytest.raises(api.TracerBoyError):
        api.DecodeImage(p)
    p2 = str(tmp_path / "short.png"); open(p2, "wb").write(src[:40])
    with pytest.raises(api.TracerBoyError):
        api.DecodeImage(p2)
    with pytest.raises(api.TracerBoyError):
        api.DecodeImage(str(tmp_path / "missing.tga"))
    p3 = str(tmp_path / "x.jpg"); open(p3, "wb").write(b"\xff\xd8\xff")
    with pytest.raises(api.TracerBoyError):
        api.DecodeImage(p3)


def test_scene_with_png_textures_loads(built, tmp_path):
    """a pbrt scene whose material uses an imagemap .png: the texel pool holds the decoded texels, the texture carries the
    texture flags CreateMaterial gives a matte Kd map and an image with transparency clears NO_ALPHA on the material"""
    from tracerboy_amd import api
    import shutil
    shutil.copy(os.path.join(IMAGES, "rgba8_smooth.png"), tmp_path / "albedo.png")
    shutil.copy(os.path.join(IMAGES, "rgb8.png"), tmp_path / "opaque.png")
    scene = '''
LookAt 0 1 5  0 1 0  0 1 0
Camera "perspective" "float fov" [40]
Film "image" "integer xresolution" [64] "integer yresolution" [48]
WorldBegin
Texture "tex-a" "spectrum" "imagemap" "string filename" ["albedo.png"]
Texture "tex-b" "spectrum" "imagemap" "string filename" ["opaque.png"]
MakeNamedMaterial "A" "string type" ["matte"] "texture Kd" ["tex-a"]
MakeNamedMaterial "B" "string type" ["matte"] "texture Kd" ["tex-b"]
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [10 10 10]
  Shape "trianglemesh" "integer indices" [0 1 2] "point P" [-1 3 -1  1 3 -1  0 3 1] "float uv" [0 0 1 0 0 1]
AttributeEnd
NamedMaterial "A"
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-2 0 -2  2 0 -2  2 0 2  -2 0 2] "float uv" [0 0 1 0 1 1 0 1]
NamedMaterial "B"
Shape "trianglemesh" "integer indices" [0 1 2] "point P" [-1 0 -1  1 0 -1  0 2 -1] "float uv" [0 0 1 0 0 1]
WorldEnd
'''
    p = tmp_path / "scene.pbrt"; p.write_text(scene)
    hs = api.HostScene(str(p))
    v = hs.view()
    assert v.numImages == 2
    a, _, _ = api.DecodeImage(str(tmp_path / "albedo.png"))
    d = v.images[0]
    import ctypes as C
    pool = np.ctypeslib.as_array(C.cast(v.texelPool, C.POINTER(C.c_float)), shape=((d.texelOffset + d.width * d.height) * 4,))
    texels = pool[d.texelOffset * 4:].reshape(-1, 4)
    assert (d.width, d.height) == (64, 40) and np.array_equal(texels, a.reshape(-1, 4))
    mats = [v.materials[i] for i in range(v.numMaterials)]
    with_alpha = [m for m in mats if m.albedoIndex != 0xffffffff and not (m.Flags & 0x20)]   # NO_ALPHA_MATERIAL_FLAG
    opaque = [m for m in mats if m.albedoIndex != 0xffffffff and (m.Flags & 0x20)]
    assert len(with_alpha) == 1 and len(opaque) == 1
    for m in with_alpha + opaque:   # matte Kd maps are created with bGammaCorrect = false (TracerBoy.cpp:429); only uber / subsurface ask for it
        assert v.textureData[m.albedoIndex].TextureFlags == 0
