"""The reference's own configs[3] scene, /root/reference/Scenes/vw-van (BASELINE.json configs[3]: 3840x2160), as far as its tree holds it:
tests/golden/scenes/vw-van = the scene file with the one absent mesh (the body shell) edited out and a synthetic sky in place of the
absent environment map, plus the 161 PLY meshes that are there -- 697 k triangles flattened, 240 ObjectInstances, glass / metal / uber /
mix materials (tests/golden/make_vw_van_fixture.py).  Real content for everything the procedural stand-ins tuned: the loader (pinned
against the reference parser's own reading of the file), the `vol` feature set (interior walks + a mix material), the two-level walk
with 240 instances, the launch policy."""
import copy
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
from conftest import GOLDEN

VW = os.path.join(GOLDEN, "scenes", "vw-van", "vw-van.pbrt")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_loader_matches_reference_parser_on_vw_van(built, tmp_path):
    """Every record the reference parser makes of the file (oracle/_ref/pbrt_dump: 143 shapes incl. the object's, their vertices / normals /
    texture coordinates / indices, 11 materials, camera, film, the infinite light's frame) -- the build's own loader reproduces all 1 425.
    This scene found what no hand-written fixture had: createMaterial_glass assigns getParam1f("index") with its fall-back of 0
    (impl/semantic/Materials.cpp:417), so a glass without an index has index 0 in the reference, not the struct's 1.5."""
    import test_host_scene as T
    mine = T.digest_records(T.loader_dump(VW, tmp_path))
    ref = json.load(open(os.path.join(GOLDEN, "vw-van.parser.digest.json")))
    assert len(mine) == len(ref) == 1425
    for a, b in zip(mine, ref):
        assert a == b, (a[:2], b[:2])


def test_vw_van_conversion(built):
    from tracerboy_amd import api
    hs = api.HostScene(VW, bvh_builder=3)
    i = hs.info()
    assert (i.numTriangles, i.numMaterials, i.numLights, i.filmWidth, i.filmHeight) == (696999, 11, 0, 1600, 1000)
    v = hs.view()
    flags = sorted(v.materials[k].Flags for k in range(i.numMaterials))
    assert sum(1 for f in flags if f & 0x2) == 4 and sum(1 for f in flags if f & 0x8) == 1   # four glass materials (interior walks), one mix material (the car paint)
    assert all(v.materials[k].IOR == 0.0 for k in range(i.numMaterials) if v.materials[k].Flags & 0x2)   # glass without "index": the reference parser's 0
    assert (v.envWidth, v.envHeight) == (256, 128)
    rc, depth = ol.validate_bvh(hs.bvh_bytes(), hs.triangles())
    assert rc == 0 and depth == i.bvhMaxDepth
    two = api.HostScene(VW, bvh_builder=3, flatten_instances=False)
    assert two.info().numTriangles == 682659      # 240 instances of the object's shapes count once


@pytest.mark.gpu
@pytest.mark.parametrize("tree", ["lbvh+treelets-gpu", "bench"])
@pytest.mark.parametrize("flatten", [1, 0])
def test_vw_van_4k_strips_bit_exact(gpu_tb, settings, flatten, tree):
    """tree = "bench": the trees bench.py's roofline_vwvan / _vwvan_2level / scale_vwvan legs time (bench.WORKLOADS imported).
    configs[3]'s frame on one GPU, flattened and as the two-level structure the reference hands its hardware path (240 instances):
    whole-frame properties + 8-row strips against the oracle (which walks the same structure)."""
    W, H, F = 3840, 2160, 2
    s = copy.copy(settings); s.MaxBounces = 6
    if tree == "bench":
        from test_gpu_parity import load_bench_workload
        w = load_bench_workload(gpu_tb, "vwvan" if flatten else "vwvan_2level")
        assert (w["W"], w["H"], w["depth"]) == (W, H, s.MaxBounces) and w["opts"]["flatten_instances"] == flatten
    else:
        gpu_tb.SetOption("flatten_instances", flatten); gpu_tb.SetOption("bvh_builder", 4)
        try:
            gpu_tb.LoadScene(VW)
        finally:
            gpu_tb.SetOption("flatten_instances", 1); gpu_tb.SetOption("bvh_builder", 0)
    gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, F, s, 0.0)
    # "vol": interior walks AND a mix material; the two-level tree is 53 levels deep: 39 entries of the tuned copy's stack in LDS (four workgroups per CU), 14 in global memory
    assert gpu_tb.GetOption("last_variant") == 3
    if tree != "bench":
        assert gpu_tb.GetOption("last_plan_stack_overflow") == (3 if flatten else 14)
    out, jit = gpu_tb.ReadAccumulation(jittered=True)
    # a sample that comes back NaN is dropped WITH its weight (RayGenCommon.h:704-727; this scene's index-0 glass makes some): weights count at most the frames
    assert not np.isnan(out).any() and (out[..., 3] <= float(F)).all() and (out[..., 3] == float(F)).mean() > 0.98 and (out[..., :3] >= 0).all() and out[..., :3].max() > 0
    view, pf = gpu_tb.HostSceneView(), gpu_tb.FrameConstants(W, H, 0, s, 0.0)
    for y0 in (700, 1100, 1500):
        ref = ol.render(view, pf, W, H, F, y0=y0, y1=y0 + 8, threads=8, jittered=True)
        assert np.array_equal(bits(out[y0:y0 + 8]), bits(ref["output"][y0:y0 + 8])), y0
        assert np.array_equal(bits(jit[y0:y0 + 8]), bits(ref["jittered"][y0:y0 + 8])), y0
    # progressive: 2 + 3 frames = 5 frames
    gpu_tb.Render(W, H, 3, s, 0.0)
    out5 = gpu_tb.ReadAccumulation()
    ref5 = ol.render(view, pf, W, H, 5, y0=1100, y1=1108, threads=8)
    assert np.array_equal(bits(out5[1100:1108]), bits(ref5["output"][1100:1108]))


@pytest.mark.gpu
def test_vw_van_two_level_equals_flattened_where_the_reference_would(gpu_tb, settings):
    """The two structures hold the same triangles in the same places; shading normals of instanced geometry stay in object space in the
    two-level scene (TracerBoy.cpp:1623-1624), so the pictures differ only where an instance is rotated -- but a camera ray's closest-hit
    distance is the same number in both.  1 000 rays through the frame."""
    rng = np.random.default_rng(5)
    hits = []
    for flatten in (1, 0):
        gpu_tb.SetOption("flatten_instances", flatten); gpu_tb.SetOption("bvh_builder", 4)
        try:
            gpu_tb.LoadScene(VW)
        finally:
            gpu_tb.SetOption("flatten_instances", 1); gpu_tb.SetOption("bvh_builder", 0)
        if flatten:
            cam = gpu_tb.GetCamera()
            o = np.tile(np.array([cam.Position[0], cam.Position[1], cam.Position[2]], np.float32), (1000, 1))
            look = np.array([cam.LookAt[0], cam.LookAt[1], cam.LookAt[2]], np.float32) - o[0]
            d = look / np.linalg.norm(look) + rng.normal(0, 0.08, (1000, 3)).astype(np.float32)
            d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        g = gpu_tb.TraceClosest(o, d)
        c = ol.trace_closest(gpu_tb.HostSceneView(), o, d)
        assert np.array_equal(bits(g["t"]), bits(c["t"])) and np.array_equal(g["prim"], c["prim"])
        hits.append(g["t"].copy())
    assert (hits[0] > 0).sum() > 300
    assert np.allclose(hits[0], hits[1], rtol=2e-5, atol=1e-3)        # object-space and world-space walks of the same geometry


def test_nan_rays_are_misses_either_way(built, tmp_path):
    """Second result-neutral deviation from the literal walk: a ray with a NaN in origin or direction cannot hit a
    triangle (the NaN reaches U, V, W of the watertight test together), but the slab test's min / max drop NaN operands, so the literal
    walk visits most of the tree before it has hit nothing -- one such ray in 12 000 of this scene took 86 % of all the steps.  Kernels
    and checker call it a miss at once; TB_LITERAL_BOX_TEST=1 walks it literally: same radiance bits, same hits, fewer boxes."""
    import subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(GOLDEN))
    code = textwrap.dedent('''
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        from tracerboy_amd import api
        import oracle_lib as ol
        s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 6
        hs = api.HostScene(%r, bvh_builder=3)
        r = ol.render(hs.view(), hs.frame_constants(s, 0, 0.0), 96, 60, 1, threads=4, stats=True)
        nan = np.float32(np.nan)
        o = np.array([[0, 100, 0]] * 4 + [[nan, 100, 0]], np.float32); d = np.array([[nan, .5, .5], [.5, nan, nan], [.3, .4, nan], [.6, 0, .8], [.6, 0, .8]], np.float32)
        t = ol.trace_closest(hs.view(), o, d)
        np.savez(sys.argv[1], r["output"], np.array([r["stats"].boxesTested, r["stats"].rays, r["stats"].hitsShaded], np.float64), t["t"], t["boxes"].astype(np.float64))
    ''') % (root, os.path.join(root, "tests"), VW)
    res = {}
    for literal in ("0", "1"):
        path = str(tmp_path / ("r%s.npz" % literal))
        subprocess.run([sys.executable, "-c", code, path], check=True, env=dict(os.environ, TB_LITERAL_BOX_TEST=literal))
        res[literal] = np.load(path)
    assert np.array_equal(bits(res["0"]["arr_0"]), bits(res["1"]["arr_0"]))                       # the picture
    assert res["0"]["arr_1"][1] == res["1"]["arr_1"][1] and res["0"]["arr_1"][2] == res["1"]["arr_1"][2]   # rays cast, hits shaded
    assert res["0"]["arr_1"][0] < 0.5 * res["1"]["arr_1"][0]                                       # boxes: the literal walk tests several times as many
    assert np.array_equal(res["0"]["arr_2"], res["1"]["arr_2"]) and (res["0"]["arr_2"][[0, 1, 2, 4]] == -1).all()   # NaN rays miss either way
    assert (res["0"]["arr_3"][[0, 1, 2, 4]] == 0).all() and res["1"]["arr_3"][[0, 1, 2]].sum() > 1000              # at once / after walking the tree


@pytest.mark.gpu
def test_ray_stats_of_nan_rays_follow_the_checker_not_the_literal_walk(gpu_tb, settings, tmp_path):
    """a ray with a NaN is a miss before the walk (ray_cannot_hit, pt_device.hpp), so the counting kernels and the heatmap
    report 0 BoxesTested / TrianglesTested for it where the reference's literal walk would count the whole tree (INTEGRATION.md,
    "Known stat divergence").  The GPU's counters equal the checker's -- rays, hits, boxes, triangles -- and the literal walk
    (TB_LITERAL_BOX_TEST=1, in a child process: the switch is read once) casts the same rays, shades the same hits and tests MORE boxes."""
    import subprocess, sys, textwrap
    W, H = 96, 60
    s = copy.copy(settings); s.MaxBounces = 6
    gpu_tb.SetOption("bvh_builder", 3)
    try:
        gpu_tb.LoadScene(VW)
    finally:
        gpu_tb.SetOption("bvh_builder", 0)
    gpu_tb.SetOption("count_rays", 1)
    try:
        gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, 1, s, 0.0)
        g = gpu_tb.ReadbackStats().rays
    finally:
        gpu_tb.SetOption("count_rays", 0)
    c = ol.render(gpu_tb.HostSceneView(), gpu_tb.FrameConstants(W, H, 0, s, 0.0), W, H, 1, threads=4, stats=True)["stats"]
    assert (g.rays, g.hitsShaded, g.boxesTested, g.trianglesTested) == (c.rays, c.hitsShaded, c.boxesTested, c.trianglesTested)
    root = os.path.dirname(os.path.dirname(GOLDEN))
    code = textwrap.dedent('''
        import sys
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        from tracerboy_amd import api
        import oracle_lib as ol
        s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 6
        hs = api.HostScene(%r, bvh_builder=3)
        st = ol.render(hs.view(), hs.frame_constants(s, 0, 0.0), %d, %d, 1, threads=4, stats=True)["stats"]
        print(st.rays, st.hitsShaded, st.boxesTested, st.trianglesTested)
    ''') % (root, os.path.join(root, "tests"), VW, W, H)
    r = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True, env=dict(os.environ, TB_LITERAL_BOX_TEST="1"))
    rays, hits, boxes, tris = (int(x) for x in r.stdout.split()[-4:])
    assert (rays, hits) == (g.rays, g.hitsShaded) and boxes > g.boxesTested and tris >= g.trianglesTested
