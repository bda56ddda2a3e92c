"""Host side of the path (no GPU): PBRT loader vs the reference parser's dumps, scene conversion,
BVH builder vs the oracle's serial restatement, BVH validator invariants (BVHValidator.cpp:60-190)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
from conftest import CORNELL, GOLDEN

TEAPOT = os.path.join(GOLDEN, "scenes", "Teapot", "scene.pbrt")


def loader_dump(path, tmp_path):
    from tracerboy_amd import api
    out = str(tmp_path / "dump.txt")
    err = C.create_string_buffer(256)
    rc = api.lib().tb_host_pbrt_dump(path.encode(), out.encode(), err, 256)
    assert rc == 0, err.value
    return out


def test_loader_matches_reference_parser_on_cornell(built, tmp_path):
    mine = open(loader_dump(CORNELL, tmp_path)).read()
    ref = open(os.path.join(GOLDEN, "cornell-box.parser.txt")).read()
    assert mine == ref  # every vertex, normal, uv, index, material parameter and the camera frame, bit for bit


def test_pbf_written_by_the_reference_parser_loads_identically(built, tmp_path):
    """`.pbf` is the reference parser's binary scene format (TracerBoy.cpp:1210-1223).  tests/golden/cornell-box.pbf was
    written by the reference's OWN parser (oracle/_ref/pbrt_dump --save-pbf); the build's independent reader must turn it
    into exactly the scene the reference parser reports for the .pbrt -- and the converted host scene must not differ."""
    from tracerboy_amd import api
    pbf = os.path.join(GOLDEN, "cornell-box.pbf")
    mine = open(loader_dump(pbf, tmp_path)).read()
    ref = open(os.path.join(GOLDEN, "cornell-box.parser.txt")).read()
    assert mine == ref
    a, b = api.HostScene(CORNELL), api.HostScene(pbf)
    assert np.array_equal(a.bvh_bytes(), b.bvh_bytes())
    ia, ib = a.info(), b.info()
    assert (ia.numTriangles, ia.numMaterials, ia.numLights, ia.filmWidth, ia.filmHeight) == (ib.numTriangles, ib.numMaterials, ib.numLights, ib.filmWidth, ib.filmHeight)
    # damaged files are rejected with an error, not a crash
    blob = open(pbf, "rb").read()
    for cut in (3, 40, len(blob) - 7):
        p = tmp_path / ("cut%d.pbf" % cut); p.write_bytes(blob[:cut])
        with pytest.raises(api.TracerBoyError):
            api.HostScene(str(p))


def digest_records(path):
    out = []
    for line in open(path):
        parts = line.rstrip("\n").split(" ")
        name, count, vals = parts[0], parts[1], parts[2:]
        if len(vals) <= 16:
            out.append([name, count, " ".join(vals)])
        else:
            out.append([name, count, "sha256:" + hashlib.sha256(" ".join(vals).encode()).hexdigest(), " ".join(vals[:6])])
    return out


def test_loader_matches_reference_parser_on_teapot(built, tmp_path):
    """126 050 triangles from two binary PLYs + checkerboard texture + infinite light transform."""
    mine = digest_records(loader_dump(TEAPOT, tmp_path))
    ref = json.load(open(os.path.join(GOLDEN, "teapot.parser.digest.json")))
    assert len(mine) == len(ref)
    for a, b in zip(mine, ref):
        if a[0] == "light_infinite":
            assert a[2].endswith("sky.hdr") and b[2].endswith("envmap.hdr")  # the only edit made to the fixture scene
            continue
        assert a == b, (a[:2], b[:2])


def test_cornell_conversion_counts(cornell_host):
    i = cornell_host.info()
    assert (i.numTriangles, i.numVertices, i.numMaterials, i.numLights, i.numGeometries) == (36, 72, 8, 2, 8)
    assert i.bvhBytesA == 16 + 32 * 71 + 52 * 36
    assert (i.filmWidth, i.filmHeight) == (800, 600)
    assert list(i.sceneMax) == [1.0, 2.0, 1.0] and abs(i.sceneMin[1] + 1.75e-7) < 1e-9
    v = cornell_host.view()
    # vertex normals are re-normalised at load (TracerBoy.cpp:1647)
    vb = np.ctypeslib.as_array(v.vertexBuffer, shape=(v.numVertexFloats,)).reshape(-1, 8)
    assert np.allclose(np.linalg.norm(vb[:, :3], axis=1), 1.0, atol=1e-6)
    assert np.all(vb[:, 5:] == [0, 0, 1])  # default tangent (TracerBoy.cpp:1644)


def test_teapot_conversion(built):
    from tracerboy_amd import api
    hs = api.HostScene(TEAPOT)
    i = hs.info()
    assert i.numTriangles == 126050 and i.numGeometries == 3 and i.numMaterials == 2 and i.numTextures == 1
    v = hs.view()
    m_floor, m_pot = v.materials[0], v.materials[1]
    assert m_floor.albedoIndex == 0 and m_floor.Flags & 0x4          # matte + checker texture
    assert abs(m_pot.SpecularCoef - 0.04) < 1e-7 and abs(m_pot.roughness - 0.001) < 1e-9 and abs(m_pot.IOR - 1.5) < 1e-6
    td = v.textureData[0]
    assert td.TextureType == 1 and td.UScale == 20.0 and td.VScale == 20.0
    assert v.envWidth == 256 and v.envHeight == 128
    cc = v.config
    assert abs(cc.EnvMapTransformVx.x + 0.386527) < 1e-6 and abs(cc.EnvMapTransformVz.y - 1.0) < 1e-6
    rc, depth = ol.validate_bvh(hs.bvh_bytes(), hs.triangles())
    assert rc == 0 and depth == i.bvhMaxDepth


@pytest.mark.parametrize("scene", ["cornell", "teapot", "proc0", "proc1", "single"])
def test_lbvh_builder_equals_oracle_restatement(built, scene, tmp_path):
    from tracerboy_amd import api
    if scene == "cornell":
        hs = api.HostScene(CORNELL)
    elif scene == "teapot":
        hs = api.HostScene(TEAPOT)
    elif scene == "proc0":
        hs = api.HostScene(procedural=(0, 20000, 1234))
    elif scene == "proc1":
        hs = api.HostScene(procedural=(1, 30000, 7))
    else:
        p = tmp_path / "one.pbrt"
        p.write_text('Camera "perspective" "float fov" [40]\nWorldBegin\nMaterial "matte"\n'
                     'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 -3 1 0 -3 0 1 -3]\nWorldEnd\n')
        hs = api.HostScene(str(p))
    tri = hs.triangles()
    mine = hs.bvh_bytes()
    ref = ol.build_lbvh(tri)
    assert np.array_equal(mine, ref)
    rc, depth = ol.validate_bvh(mine, tri)
    assert rc == 0 and depth == hs.info().bvhMaxDepth


@pytest.mark.parametrize("scene", ["cornell", "teapot", "proc0", "proc1", "six", "seven"])
def test_treelet_builder_equals_oracle_restatement(built, scene, tmp_path):
    """Builder 3 = LBVH + the fallback layer's three treelet passes (what a PREFER_FAST_TRACE build, TracerBoy.cpp:1970, gives
    the software traversal) against the serial restatement of TreeletReorder.hlsl / FindTreelets.hlsl in oracle/bvh_ref.cpp."""
    from tracerboy_amd import api
    if scene == "cornell": mk = lambda b: api.HostScene(CORNELL, bvh_builder=b)
    elif scene == "teapot": mk = lambda b: api.HostScene(TEAPOT, bvh_builder=b)
    elif scene == "proc0": mk = lambda b: api.HostScene(procedural=(0, 20000, 1234), bvh_builder=b)
    elif scene == "proc1": mk = lambda b: api.HostScene(procedural=(1, 30000, 7), bvh_builder=b)
    else:
        n = 6 if scene == "six" else 7   # below / at FullTreeletSize: no pass / exactly one treelet at the root
        rng = np.random.default_rng(n)
        shapes = "\n".join('Shape "trianglemesh" "integer indices" [0 1 2] "point P" [%s]' % " ".join("%.4f" % v for v in rng.uniform(-1, 1, 9) + [0, 0, -4] * 3) for _ in range(n))
        p = tmp_path / "few.pbrt"
        p.write_text('Camera "perspective" "float fov" [40]\nWorldBegin\nMaterial "matte"\n' + shapes + "\nWorldEnd\n")
        mk = lambda b: api.HostScene(str(p), bvh_builder=b)
    hs, plain = mk(3), mk(0)
    tri = hs.triangles()
    mine = hs.bvh_bytes()
    assert np.array_equal(mine, ol.build_lbvh(tri, 3))
    assert np.array_equal(mine, plain.bvh_bytes()) == (scene == "six")
    rc, depth = ol.validate_bvh(mine, tri)
    assert rc == 0 and depth == hs.info().bvhMaxDepth
    if scene in ("cornell", "proc0"):
        s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 3
        ra = ol.render(plain.view(), plain.frame_constants(s), 48, 32, 1, stats=True)
        rb = ol.render(hs.view(), hs.frame_constants(s), 48, 32, 1, stats=True)
        assert rb["stats"].boxesTested < 0.7 * ra["stats"].boxesTested
        assert np.array_equal(ra["output"], rb["output"])


def test_treelet_subset_tables_are_the_reference_tables():
    """Known answers held by the reference (TreeletReorderBindings.h:63-73): C(7, k) and the table of 7-bit masks with k bits
    set that FindOptimalPartitions walks per subset size -- the oracle and the builders walk every mask of that popcount, so
    the two must be the same sets; the split enumeration (delta / partition trick) must visit every split of a mask once,
    always keeping the lowest leaf on the far side."""
    choose = [1, 7, 21, 35, 35, 21, 7, 1]
    rows = {2: [0x03, 0x05, 0x06, 0x09, 0x0a, 0x0c, 0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x30, 0x41, 0x42, 0x44, 0x48, 0x50, 0x60],
            3: [0x07, 0x0b, 0x0d, 0x0e, 0x13, 0x15, 0x16, 0x19, 0x1a, 0x1c, 0x23, 0x25, 0x26, 0x29, 0x2a, 0x2c, 0x31, 0x32, 0x34, 0x38, 0x43, 0x45, 0x46, 0x49, 0x4a,
                0x4c, 0x51, 0x52, 0x54, 0x58, 0x61, 0x62, 0x64, 0x68, 0x70],
            4: [0x0f, 0x17, 0x1b, 0x1d, 0x1e, 0x27, 0x2b, 0x2d, 0x2e, 0x33, 0x35, 0x36, 0x39, 0x3a, 0x3c, 0x47, 0x4b, 0x4d, 0x4e, 0x53, 0x55, 0x56, 0x59, 0x5a, 0x5c,
                0x63, 0x65, 0x66, 0x69, 0x6a, 0x6c, 0x71, 0x72, 0x74, 0x78],
            5: [0x1f, 0x2f, 0x37, 0x3b, 0x3d, 0x3e, 0x4f, 0x57, 0x5b, 0x5d, 0x5e, 0x67, 0x6b, 0x6d, 0x6e, 0x73, 0x75, 0x76, 0x79, 0x7a, 0x7c],
            6: [0x3f, 0x5f, 0x6f, 0x77, 0x7b, 0x7d, 0x7e]}
    for k in range(8):
        masks = [m for m in range(128) if bin(m).count("1") == k]
        assert len(masks) == choose[k]
        if k in rows: assert masks == rows[k]
    for m in range(1, 128):
        if bin(m).count("1") < 2: continue
        delta = (m - 1) & m; p = (-delta) & m; seen = []
        while True:
            seen.append(p); p = (p - delta) & m
            if p == 0: break
        low = m & -m
        assert len(set(seen)) == len(seen) == 2 ** (bin(m).count("1") - 1) - 1
        assert all(q & m == q and q and not (q & low) for q in seen)


def test_sah_builder_is_valid_and_cheaper(built):
    from tracerboy_amd import api
    a = api.HostScene(CORNELL, bvh_builder=0)
    b = api.HostScene(CORNELL, bvh_builder=1)
    tri = b.triangles()
    rc, _ = ol.validate_bvh(b.bvh_bytes(), tri)
    assert rc == 0
    s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 3
    ra = ol.render(a.view(), a.frame_constants(s), 32, 32, 1, stats=True)
    rb = ol.render(b.view(), b.frame_constants(s), 32, 32, 1, stats=True)
    assert rb["stats"].boxesTested < ra["stats"].boxesTested
    # a different but valid tree finds the same surfaces: identical first-hit depth buffer up to ties
    o = np.array([[0.0, 1.0, 6.79]] * 64, np.float32)
    d = np.stack([np.linspace(-0.15, 0.15, 64), np.linspace(0.1, -0.1, 64), -np.ones(64)], 1).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ha, hb = ol.trace_closest(a.view(), o, d), ol.trace_closest(b.view(), o, d)
    assert np.array_equal(ha["t"], hb["t"])


def test_reinsertion_passes_keep_the_tree_valid_and_cut_box_tests(built, monkeypatch):
    """The SAH builder's reinsertion passes (bvh_build.cpp, optimizeByReinsertion) only move subtrees: the result is still
    a tree the reference's validator accepts, it is no deeper than the traversal stack allows, finds the same surfaces,
    and a path-traced frame tests fewer boxes than with the passes turned off."""
    from tracerboy_amd import api
    s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 3
    for make in (lambda **kw: api.HostScene(CORNELL, bvh_builder=1, **kw), lambda **kw: api.HostScene(procedural=(0, 6000, 1234), bvh_builder=1, **kw)):
        plain = make(reinsertion_passes=0)
        monkeypatch.setenv("TB_REINSERT_PASSES", "0")    # an environment variable no longer reaches the builder
        opt = make()
        tri = opt.triangles()
        assert np.array_equal(tri["positions"], plain.triangles()["positions"])
        rc, depth = ol.validate_bvh(opt.bvh_bytes(), tri)
        assert rc == 0 and depth == opt.info().bvhMaxDepth and depth <= 35 + 1
        assert not np.array_equal(opt.bvh_bytes(), plain.bvh_bytes())
        ra = ol.render(plain.view(), plain.frame_constants(s), 48, 32, 1, stats=True)
        rb = ol.render(opt.view(), opt.frame_constants(s), 48, 32, 1, stats=True)
        assert rb["stats"].boxesTested < ra["stats"].boxesTested
        assert np.array_equal(ra["output"], rb["output"])


def test_layout_b_is_the_same_tree_as_layout_a(cornell_host):
    nodes, tris, root = cornell_host.layout_b()
    bvh = cornell_host.bvh_bytes()
    n = cornell_host.info().numTriangles
    a_nodes = bvh[16:16 + 32 * (2 * n - 1)].view(np.uint32).reshape(-1, 8)
    assert root == 0 and nodes.shape[0] == n - 1 and tris.shape[0] == n
    for i in range(n - 1):
        l, r = a_nodes[i, 3] & 0xffffff, a_nodes[i, 7]
        # layout B interleaves the two child boxes per component: cx[2] cy[2] cz[2] hx[2] hy[2] hz[2] left right
        for k, child in enumerate((l, r)):
            assert np.array_equal(nodes[i, [0 + k, 2 + k, 4 + k]], a_nodes[child, 0:3])     # centre
            assert np.array_equal(nodes[i, [6 + k, 8 + k, 10 + k]], a_nodes[child, 4:7])    # half-extent
        for ref, child in ((nodes[i, 12], l), (nodes[i, 13], r)):
            if child >= n - 1:
                assert ref == (0x80000000 | (child - (n - 1)))
            else:
                assert ref == child


def test_blue_noise_tiles_are_the_reference_tiles(cornell_host):
    v = cornell_host.view()
    got = np.ctypeslib.as_array(C.cast(v.blueNoise0, C.POINTER(C.c_float)), shape=(256 * 256 * 4,))
    raw = np.fromfile(os.path.join(GOLDEN, "bluenoise0.rgba8"), np.uint8).astype(np.float32) / np.float32(255.0)
    assert np.array_equal(got, raw)


def test_tile_unpack_roundtrip(built):
    from tracerboy_amd import api
    W, H, world, tw, th = 100, 70, 3, 32, 16
    full = np.random.default_rng(0).random((H, W, 4), np.float32)
    tilesX, tilesY = -(-W // tw), -(-H // th)
    packed = [np.zeros((-(-(tilesX * tilesY - r) // world) * tw * th, 4), np.float32) for r in range(world)]
    for t in range(tilesX * tilesY):
        r, local = t % world, t // world
        x0, y0 = (t % tilesX) * tw, (t // tilesX) * th
        blk = full[y0:y0 + th, x0:x0 + tw]
        packed[r][local * tw * th: local * tw * th + blk.shape[0] * blk.shape[1]] = blk.reshape(-1, 4)
    out = api.unpack_gathered(W, H, world, tw, th, packed)
    assert np.array_equal(out, full)


def test_degenerate_axis_culling_does_not_change_radiance(built, tmp_path):
    """The build culls boxes on axes where the ray direction is exactly 0 (the reference's slab test lets every
    such box through, TraverseFunction.hlsli:212-214).  Radiance must be bit-identical to the literal test;
    only the box counters may drop."""
    import subprocess, sys, textwrap
    code = textwrap.dedent('''
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        from tracerboy_amd import api
        import oracle_lib as ol
        s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 6
        out = []
        for hs in (api.HostScene(%r), api.HostScene(procedural=(0, 6000, 5))):
            r = ol.render(hs.view(), hs.frame_constants(s, 0, 0.0), 96, 64, 3, threads=4, stats=True)
            out.append(r["output"]); out.append(np.array([r["stats"].boxesTested, r["stats"].trianglesTested], np.float64))
        np.savez(sys.argv[1], *out)
    ''') % (os.path.dirname(GOLDEN.rstrip("/")).rsplit("/tests", 1)[0], os.path.dirname(GOLDEN), CORNELL)
    res = {}
    for literal in ("0", "1"):
        path = str(tmp_path / ("r%s.npz" % literal))
        env = dict(os.environ, TB_LITERAL_BOX_TEST=literal)
        subprocess.run([sys.executable, "-c", code, path], check=True, env=env)
        res[literal] = np.load(path)
    for k in ("arr_0", "arr_2"):
        assert np.array_equal(res["0"][k].view(np.uint32), res["1"][k].view(np.uint32))
    assert res["0"]["arr_1"][0] < res["1"]["arr_1"][0]       # fewer boxes on cornell (axis-aligned walls, rand() == 0 happens)
    assert res["0"]["arr_3"][0] < 0.8 * res["1"]["arr_3"][0]  # far fewer on the tessellated blob


def test_presplit_references_keep_the_picture_and_the_invariants(built):
    """Option presplit (bvh_build.cpp presplitReferences, round 6): the SAH builder may cut the triangles with the largest, emptiest boxes into
    parts before it builds.  Leaves then outnumber triangles; every leaf still holds a whole input triangle, a part's box lies inside its triangle's
    bounds, the parts' boxes cover the triangle (oracle/bvh_ref.cpp validates that form too), closest hits and the path-traced picture are the same
    bits as without -- and triangle tests per sample fall while box tests RISE, which is why no workload uses it (docs/experiments/r6.md)."""
    from tracerboy_amd import api
    s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 4
    for kw in (dict(path=os.path.join(GOLDEN, "scenes", "Teapot", "scene.pbrt")), dict(procedural=(1, 20000, 7))):
        plain = api.HostScene(bvh_builder=1, reinsertion_passes=1, **kw)
        cut = api.HostScene(bvh_builder=1, reinsertion_passes=1, presplit=30, **kw)
        n = plain.info().numTriangles
        assert cut.info().numTriangles == n
        leaves_plain, leaves_cut = (plain.view().bvhBytes - 16 + 32) // 116, (cut.view().bvhBytes - 16 + 32) // 116
        assert leaves_plain == n and n < leaves_cut <= n + n * 30 // 100
        rc, depth = ol.validate_bvh(cut.bvh_bytes(), cut.triangles())
        assert rc == 0 and depth == cut.info().bvhMaxDepth
        ra = ol.render(plain.view(), plain.frame_constants(s), 64, 40, 2, threads=4, stats=True)
        rb = ol.render(cut.view(), cut.frame_constants(s), 64, 40, 2, threads=4, stats=True)
        assert np.array_equal(ra["output"].view(np.uint32), rb["output"].view(np.uint32))
        assert rb["stats"].trianglesTested < ra["stats"].trianglesTested
    # a tree over parts with one part's box shrunk no longer covers its triangle: the validator says so
    cut = api.HostScene(procedural=(1, 20000, 7), bvh_builder=1, reinsertion_passes=0, presplit=30)
    img = np.frombuffer(cut.bvh_bytes(), np.uint8).copy()
    tri = cut.triangles()
    leaves = (img.size - 16 + 32) // 116
    nodes = img[16:16 + 32 * (2 * leaves - 1)].view(np.float32).reshape(-1, 8)
    flags = img[16:16 + 32 * (2 * leaves - 1)].view(np.uint32).reshape(-1, 8)
    prims = img[16 + 32 * (2 * leaves - 1):16 + 32 * (2 * leaves - 1) + 40 * leaves].reshape(leaves, 40)
    verts = prims[:, 4:40].copy().view(np.float32).reshape(leaves, 9)
    key = [v.tobytes() for v in verts]
    import collections
    dup = [k for k, c in collections.Counter(key).items() if c > 1]
    assert dup, "no triangle was cut"
    leaf_nodes = np.nonzero(flags[:, 3] & 0x80000000)[0]
    victim = next(i for i in leaf_nodes if key[int(flags[i, 3] & 0x00ffffff)] == dup[0])
    nodes[victim, 4:7] *= 0.25                                  # halfDim of that part
    assert ol.validate_bvh(img, tri)[0] in (-12, -9)
