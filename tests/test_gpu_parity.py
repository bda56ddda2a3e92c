"""-m gpu: the HIP path against the CPU oracle through the C ABI.  fp32 radiance, bar = BIT-EXACT
(stronger than the 1e-4 relative L2 BASELINE.json asks for); integer outputs (hit indices, counters) exact."""
import copy
import os

import numpy as np
import pytest

import oracle_lib as ol
from conftest import CORNELL, GOLDEN

pytestmark = pytest.mark.gpu
TEAPOT = os.path.join(GOLDEN, "scenes", "Teapot", "scene.pbrt")
MIX_GLASS = os.path.join(GOLDEN, "scenes", "mix-glass", "scene.pbrt")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def rel_l2(a, b):
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


def test_native_library_is_the_code_under_test(gpu_tb):
    from tracerboy_amd import api
    assert os.path.basename(api.LIB_PATH) == "libtracerboy_hip.so" and os.path.exists(api.LIB_PATH)
    maps = open("/proc/self/maps").read()
    assert "libtracerboy_hip.so" in maps


def test_device_math_is_bit_identical_to_host(gpu_tb):
    """tb_math.h evaluated by gfx950 == evaluated by the host compiler, for every function the path uses."""
    rng = np.random.default_rng(1)
    n = 200000
    cases = {
        0: (rng.uniform(-1000, 1000, n), None), 1: (rng.uniform(-1000, 1000, n), None),
        2: (rng.uniform(-1, 1, n), None), 10: (rng.uniform(-1, 1, n), None),
        3: (rng.normal(0, 3, n), rng.normal(0, 3, n)),
        4: (rng.uniform(-90, 90, n), None), 5: (np.exp(rng.uniform(-80, 80, n)), None),
        6: (rng.uniform(0, 4, n), rng.uniform(-8, 8, n)), 7: (np.exp(rng.uniform(-80, 80, n)), None),
        8: (rng.uniform(-150, 130, n), None), 9: (np.exp(rng.uniform(-100, 88, n)), None),
    }
    edge = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, np.inf, -np.inf, np.nan, 1e-45, 1e-38, 3.4e38, 1e9, 2.5e9, 6.2831855], np.float32)
    L = ol.lib()
    for fn, (a, b) in cases.items():
        a = np.concatenate([a.astype(np.float32), edge]); b2 = None
        if b is not None:
            b2 = np.concatenate([b.astype(np.float32), edge[::-1]])
        dev = gpu_tb.DeviceMath(fn, a, b2)
        sub = np.concatenate([np.arange(0, 3000), np.arange(a.size - edge.size, a.size)])
        host = np.array([L.tbo_math(fn, float(a[i]), float(b2[i]) if b2 is not None else 0.0) for i in sub], np.float32)
        d, h = bits(dev[sub]), bits(host)
        same = (d == h) | (np.isnan(dev[sub]) & np.isnan(host))
        assert same.all(), (fn, a[sub][~same][:5], dev[sub][~same][:5], host[~same][:5])
    # division, the RNG expression and hash13 (codes 11-13) including denormal results
    a = np.concatenate([rng.uniform(-1e3, 1e3, 5000), [1e-38, 3e-39, 1.0]]).astype(np.float32)
    b = np.concatenate([rng.uniform(-1e3, 1e3, 5000), [3.0, 7.0, 3.0]]).astype(np.float32)
    assert np.array_equal(bits(gpu_tb.DeviceMath(11, a, b)), bits(a / b))
    s = rng.uniform(0, 300, 4000).astype(np.float32); t = rng.uniform(0, 1, 4000).astype(np.float32)
    host = np.empty(4000, np.float32)
    import ctypes as C
    for i in range(4000):
        o = np.zeros(1, np.float32); L.tbo_rand_stream(float(s[i]), float(t[i]), 1, o.ctypes.data_as(C.c_void_p)); host[i] = o[0]
    assert np.array_equal(bits(gpu_tb.DeviceMath(12, s, t)), bits(host))
    x = rng.integers(0, 4096, 4000).astype(np.float32); y = rng.integers(0, 4096, 4000).astype(np.float32)
    host = np.array([L.tbo_hash13(float(x[i]), float(y[i]), 0.0) for i in range(4000)], np.float32)
    assert np.array_equal(bits(gpu_tb.DeviceMath(13, x, y)), bits(host))


def _camera_rays(tb, W, H, s):
    import ctypes as C
    pf = tb.FrameConstants(W, H, 0, s, 0.0)
    lens = tb.GetCamera().LensHeight
    o = np.zeros((W * H, 3), np.float32); d = np.zeros((W * H, 3), np.float32)
    oo = (C.c_float * 3)(); dd = (C.c_float * 3)()
    k = 0
    for y in range(H):
        for x in range(W):
            ol.lib().tbo_camera_ray(C.byref(pf), lens, W, H, x + 0.5, y + 0.5, 0.3, 0.7, C.byref(oo), C.byref(dd))
            o[k] = oo[:]; d[k] = dd[:]; k += 1
    return o, d


@pytest.mark.parametrize("scene", ["cornell", "teapot", "proc"])
def test_trace_closest_matches_oracle_exactly(gpu_tb, settings, scene):
    """Traverse + GetHitInfo: t, barycentrics, primitive/geometry index, interpolated normal, uv AND the
    reference's BoxesTested / TrianglesTested counters (TraverseFunction.hlsli:662,751)."""
    if scene == "cornell":
        gpu_tb.LoadScene(CORNELL)
    elif scene == "teapot":
        gpu_tb.LoadScene(TEAPOT)
    else:
        gpu_tb.LoadProcedural(0, 60000, 1234)
    view = gpu_tb.HostSceneView()
    o, d = _camera_rays(gpu_tb, 48, 32, settings)
    rng = np.random.default_rng(3)
    info = gpu_tb.SceneInfo()
    lo, hi = np.array(info.sceneMin[:]), np.array(info.sceneMax[:])
    ro = rng.uniform(lo, hi, (3000, 3)).astype(np.float32)
    rd = rng.normal(size=(3000, 3)).astype(np.float32); rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    axis = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 200)] * rng.choice([-1, 1], 200)[:, None]  # zero components -> inf in 1/d
    O = np.concatenate([o, ro, ro[:200]]); D = np.concatenate([d, rd, axis.astype(np.float32)])
    g = gpu_tb.TraceClosest(O, D)
    c = ol.trace_closest(view, O, D)
    assert (c["t"] > 0).sum() > 1000
    for k in ("t", "bary", "normal", "uv"):
        assert np.array_equal(bits(g[k]), bits(c[k])), k
    for k in ("material", "prim", "geom", "boxes", "tris"):
        assert np.array_equal(g[k], c[k]), k


def load_bench_workload(tb, key):
    """Load workload `key` of bench.WORKLOADS the way bench.py does -- the dict is imported, not restated: its builder and its
    tb_set_option()s (reinsertion passes / share, flatten_instances) -- so that the full-size parity tests walk the very trees the
    bench times (the tests' trees were not the bench's).  Options outlive a load: back to the defaults afterwards."""
    import bench
    w = bench.WORKLOADS[key]
    tb.SetOption("bvh_builder", w["builder"]); tb.SetOption("reinsertion_passes", -1); tb.SetOption("reinsertion_share", 100)
    try:
        for k, v in (w.get("opts") or {}).items():
            tb.SetOption(k, v)
        scene = w["scene"]
        if scene == "cornell-box":
            tb.LoadScene(bench.CORNELL)
        elif scene.startswith("proc"):
            kind, tris = scene[4:].split(":")
            tb.LoadProcedural(int(kind), int(tris), 1234)
        else:
            tb.LoadScene(scene)
    finally:
        tb.SetOption("bvh_builder", 0); tb.SetOption("reinsertion_passes", -1); tb.SetOption("reinsertion_share", 100)
        tb.SetOption("flatten_instances", 1)
    return w


def _oracle(tb, W, H, frames, s, **kw):
    return ol.render(tb.HostSceneView(), tb.FrameConstants(W, H, 0, s, 0.0), W, H, frames, threads=8, **kw)


def test_cornell_radiance_is_bit_exact(gpu_tb, settings):
    gpu_tb.LoadScene(CORNELL)
    W, H, F = 160, 96, 4
    gpu_tb.Render(W, H, F, settings, 0.0)
    out, jit = gpu_tb.ReadAccumulation(jittered=True)
    ref = _oracle(gpu_tb, W, H, F, settings, jittered=True)
    assert rel_l2(out, ref["output"]) <= 1e-4  # the bar BASELINE.json states
    assert np.array_equal(bits(out), bits(ref["output"]))  # the bar this build holds itself to
    assert np.array_equal(bits(jit), bits(ref["jittered"]))
    assert gpu_tb.GetNumberOfSamplesSinceLastInvalidate() == F
    # golden regression of the oracle image itself (committed fixture)
    gpu_tb.Render(64, 48, 3, settings, 0.0)
    gold = np.load(os.path.join(GOLDEN, "cornell_oracle_64x48x3.npy"))
    assert np.array_equal(bits(gpu_tb.ReadAccumulation()), bits(gold))


def test_config_c1_cornell_512_4spp_depth4(gpu_tb, settings):
    """BASELINE.json configs[0]: 512x512, 4 spp, max depth 4 -- the reference's CPU-runnable case, in full."""
    gpu_tb.LoadScene(CORNELL)
    gpu_tb.Render(512, 512, 4, settings, 0.0)
    out = gpu_tb.ReadAccumulation()
    ref = _oracle(gpu_tb, 512, 512, 4, settings)["output"]
    assert np.array_equal(bits(out), bits(ref))


def test_progressive_accumulation_and_invalidate(gpu_tb, settings):
    gpu_tb.LoadScene(CORNELL)
    W, H = 80, 48
    gpu_tb.Render(W, H, 5, settings, 0.0)
    one = gpu_tb.ReadAccumulation()
    gpu_tb.InvalidateHistory()
    gpu_tb.Render(W, H, 2, settings, 0.0)
    gpu_tb.Render(W, H, 3, settings, 0.0)  # frames 2..4 accumulate on top (RayGenCommon.h:721-722)
    assert gpu_tb.GetNumberOfSamplesSinceLastInvalidate() == 5
    assert np.array_equal(bits(gpu_tb.ReadAccumulation()), bits(one))
    s2 = copy.copy(settings); s2.MaxBounces = 2
    gpu_tb.Render(W, H, 1, s2, 0.0)  # a history-relevant setting restarts accumulation (TracerBoy.cpp:2163-2185)
    assert gpu_tb.GetNumberOfSamplesSinceLastInvalidate() == 1
    assert np.array_equal(bits(gpu_tb.ReadAccumulation()), bits(_oracle(gpu_tb, W, H, 1, s2)["output"]))


def test_time_seed_changes_the_stream_identically(gpu_tb, settings):
    gpu_tb.LoadScene(CORNELL)
    W, H = 64, 40
    gpu_tb.Render(W, H, 2, settings, 0.37)
    ref = ol.render(gpu_tb.HostSceneView(), gpu_tb.FrameConstants(W, H, 0, settings, 0.37), W, H, 2, threads=4)["output"]
    assert np.array_equal(bits(gpu_tb.ReadAccumulation()), bits(ref))


@pytest.mark.parametrize("variant", ["blue_noise", "triangle_filter", "gaussian_filter", "dof", "firefly", "ris", "no_nee", "realtime", "depth1", "depth0"])
def test_settings_variants_bit_exact(gpu_tb, settings, variant):
    gpu_tb.LoadScene(CORNELL)
    s = copy.copy(settings)
    if variant == "blue_noise": s.EnableBlueNoise = 1
    if variant == "triangle_filter": s.FilterType = 1
    if variant == "gaussian_filter": s.FilterType = 2; s.FilterWidth = 1.5
    if variant == "dof": s.DOFFocalDistance = 6.0; s.ApertureWidth = 0.05
    if variant == "firefly": s.FireflyClampValue = 2.0
    if variant == "ris": s.EnableSamplingImportanceResampling = 1
    if variant == "no_nee": s.EnableNextEventEstimation = 0
    if variant == "realtime": s.RenderModeRealTime = 1
    if variant == "depth1": s.MaxBounces = 1
    if variant == "depth0": s.MaxBounces = 0
    W, H, F = 72, 40, 3
    gpu_tb.Render(W, H, F, s, 0.0)
    out, jit = gpu_tb.ReadAccumulation(jittered=True)
    ref = _oracle(gpu_tb, W, H, F, s, jittered=True)
    assert np.array_equal(bits(out), bits(ref["output"])), variant
    assert np.array_equal(bits(jit), bits(ref["jittered"])), variant
    if variant in ("blue_noise", "no_nee", "depth1", "depth0"):   # settings every pipeline's kernels carry (no FEAT_EXT needed)
        try:
            for pipeline in (1, 2, 3):
                gpu_tb.SetOption("pipeline", pipeline); gpu_tb.InvalidateHistory()
                gpu_tb.Render(W, H, F, s, 0.0)
                assert np.array_equal(bits(gpu_tb.ReadAccumulation()), bits(ref["output"])), (variant, pipeline)
        finally:
            gpu_tb.SetOption("pipeline", 0)


def test_aovs_and_heatmap_match(gpu_tb, settings):
    gpu_tb.LoadScene(CORNELL)
    gpu_tb.SetOption("aov", 1)
    try:
        W, H = 64, 40
        gpu_tb.Render(W, H, 2, settings, 0.0)
        ref = _oracle(gpu_tb, W, H, 2, settings, aovs=True)
        for which, key in ((2, "normals"), (3, "worldpos0"), (4, "worldpos1"), (5, "custom"), (6, "depth"), (7, "emissive")):
            assert np.array_equal(bits(gpu_tb.ReadAOV(which)), bits(ref[key])), key
        s = copy.copy(settings); s.OutputType = 9  # heatmap: (TrianglesTested, BoxesTested) of the primary ray
        gpu_tb.Render(W, H, 1, s, 0.0)
        ref = _oracle(gpu_tb, W, H, 1, s, aovs=True)
        assert np.array_equal(bits(gpu_tb.ReadAOV(5)), bits(ref["custom"]))
    finally:
        gpu_tb.SetOption("aov", 0)


def test_ray_counters_equal_oracle(gpu_tb, settings):
    """The byte model's event counts come from the kernels themselves and equal the oracle's."""
    gpu_tb.LoadScene(CORNELL)
    gpu_tb.SetOption("count_rays", 1)
    try:
        W, H, F = 96, 64, 2
        gpu_tb.Render(W, H, F, settings, 0.0)
        st = gpu_tb.ReadbackStats().rays
        ref = _oracle(gpu_tb, W, H, F, settings, stats=True)["stats"]
        for k in ("boxesTested", "trianglesTested", "hitsShaded", "materialFetches", "lightSamples", "samples", "rays"):
            assert getattr(st, k) == getattr(ref, k), k
        assert np.array_equal(bits(gpu_tb.ReadAccumulation()), bits(_oracle(gpu_tb, W, H, F, settings)["output"]))
    finally:
        gpu_tb.SetOption("count_rays", 0)


def test_teapot_textures_env_specular_bit_exact(gpu_tb, settings):
    """Checker texture, RGBE environment map (atan2/acos lookup), substrate (GGX) BSDF: 126 050 triangles."""
    gpu_tb.LoadScene(TEAPOT)
    W, H, F = 96, 54, 2
    s = copy.copy(settings); s.MaxBounces = 5
    gpu_tb.Render(W, H, F, s, 0.0)
    ref = _oracle(gpu_tb, W, H, F, s)["output"]
    out = gpu_tb.ReadAccumulation()
    assert not np.isnan(out).any() and out[..., :3].max() > 0
    assert np.array_equal(bits(out), bits(ref))


@pytest.mark.parametrize("kind", [1, 2])
def test_procedural_glass_mirror_plastic_bit_exact(gpu_tb, settings, kind):
    """SSS interior random walk (refraction / TIR), mirrors, metals, plastics, many materials."""
    gpu_tb.LoadProcedural(kind, 40000, 99)
    W, H, F = 80, 48, 2
    s = copy.copy(settings); s.MaxBounces = 6
    gpu_tb.Render(W, H, F, s, 0.0)
    ref = _oracle(gpu_tb, W, H, F, s)["output"]
    assert np.array_equal(bits(gpu_tb.ReadAccumulation()), bits(ref))


def test_presplit_tree_bit_exact(gpu_tb, settings):
    """Option presplit (round 6): a tree whose leaves outnumber the triangles (the largest, emptiest triangle boxes cut into parts; a part's leaf
    holds the whole triangle) -- the kernels walk it like any other, the picture is the oracle's on the same image, and the same bits as the tree
    without."""
    W, H, F = 96, 64, 3
    s = copy.copy(settings); s.MaxBounces = 6
    pics = []
    for pre in (0, 30):
        gpu_tb.SetOption("bvh_builder", 1); gpu_tb.SetOption("presplit", pre)
        try:
            gpu_tb.LoadProcedural(1, 40000, 99)
        finally:
            gpu_tb.SetOption("bvh_builder", 0); gpu_tb.SetOption("presplit", 0)
        gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, F, s, 0.0)
        pics.append(gpu_tb.ReadAccumulation())
        assert np.array_equal(bits(pics[-1]), bits(_oracle(gpu_tb, W, H, F, s)["output"]))
        leaves = (gpu_tb.HostSceneView().bvhBytes - 16 + 32) // 116
        assert (leaves > gpu_tb.SceneInfo().numTriangles) == (pre > 0)
    assert np.array_equal(bits(pics[0]), bits(pics[1]))


def test_all_kernel_variants_agree(gpu_tb, settings):
    """Feature-stripped kernel variants only remove branches the scene can never take."""
    gpu_tb.LoadScene(CORNELL)
    W, H, F = 64, 40, 3
    gpu_tb.Render(W, H, F, settings, 0.0)
    assert gpu_tb.GetOption("last_variant") == 0  # "matte"
    a = gpu_tb.ReadAccumulation()
    gpu_tb.SetOption("force_full_variant", 1); gpu_tb.InvalidateHistory()
    try:
        gpu_tb.Render(W, H, F, settings, 0.0)
        assert gpu_tb.GetOption("last_variant") == 4
        assert np.array_equal(bits(a), bits(gpu_tb.ReadAccumulation()))
    finally:
        gpu_tb.SetOption("force_full_variant", 0)
    # the two schedulings of the same step functions (streaming vs lock-step bounce) agree as well
    for pipeline in (0, 1, 2, 3):
        gpu_tb.SetOption("pipeline", pipeline); gpu_tb.InvalidateHistory()
        gpu_tb.Render(W, H, F, settings, 0.0)
        assert np.array_equal(bits(a), bits(gpu_tb.ReadAccumulation())), pipeline
    gpu_tb.SetOption("pipeline", 0)


@pytest.mark.parametrize("scene", ["cornell", "proc0", "proc1", "mix-glass"])
def test_occupancy_copies_agree(gpu_tb, settings, scene):
    """The matte / env / vol feature sets exist twice: at the occupancy their registers allow and held to one more wave per SIMD
    (pt_variant_{matte5,env5,vol4}.hip, picked when LDS has room for that many workgroups per CU; option high_occupancy).  Both
    copies, in the one-pixel-per-lane form (3 frames) and the frame-group form (9 frames), against the oracle."""
    if scene == "cornell": gpu_tb.LoadScene(CORNELL); variant = 0
    elif scene == "proc0": gpu_tb.LoadProcedural(0, 20000, 3); variant = 1
    elif scene == "proc1": gpu_tb.LoadProcedural(1, 30000, 7); variant = 5
    else: gpu_tb.LoadScene(MIX_GLASS); variant = 3
    W, H = 96, 64
    s = copy.copy(settings); s.MaxBounces = 5
    try:
        for frames in (3, 9):
            ref = None
            gpu_tb.SetOption("frame_group", -1 if frames == 3 else 0)
            for high in (1, 0):
                gpu_tb.SetOption("high_occupancy", high); gpu_tb.InvalidateHistory()
                gpu_tb.Render(W, H, frames, s, 0.0)
                assert gpu_tb.GetOption("last_variant") == variant
                out = gpu_tb.ReadAccumulation()
                if ref is None: ref = _oracle(gpu_tb, W, H, frames, s)["output"]
                assert np.array_equal(bits(out), bits(ref)), (frames, high)
    finally:
        gpu_tb.SetOption("high_occupancy", 1); gpu_tb.SetOption("frame_group", 0)


@pytest.mark.parametrize("scene", ["cornell", "proc0", "proc1", "mix-glass"])
def test_split_traversal_stack_bit_exact(gpu_tb, settings, scene):
    """Trees too deep for the LDS share of a higher-occupancy kernel copy keep the first entries of the traversal stack in LDS
    and the deepest ones in global memory (HYBRID kernels, frame-group launches).  Forced here with a tiny LDS part
    (stack_lds_cap = 3, so that nearly every ray overflows) on the matte (scene image in LDS), env and vol feature sets, against
    the oracle."""
    if scene == "cornell": gpu_tb.LoadScene(CORNELL); variant = 0
    elif scene == "proc0": gpu_tb.LoadProcedural(0, 20000, 3); variant = 1
    elif scene == "proc1": gpu_tb.LoadProcedural(1, 30000, 7); variant = 5
    else: gpu_tb.LoadScene(MIX_GLASS); variant = 3
    W, H, F = 96, 64, 9
    s = copy.copy(settings); s.MaxBounces = 5
    try:
        gpu_tb.SetOption("stack_lds_cap", 3); gpu_tb.SetOption("stack_overflow_max", 64)
        gpu_tb.Render(W, H, F, s, 0.0)
        assert gpu_tb.GetOption("last_variant") == variant
        out = gpu_tb.ReadAccumulation()
        ref = _oracle(gpu_tb, W, H, F, s)["output"]
        assert np.array_equal(bits(out), bits(ref))
    finally:
        gpu_tb.SetOption("stack_lds_cap", 0); gpu_tb.SetOption("stack_overflow_max", 24)


@pytest.mark.parametrize("sort", [0, 1])
@pytest.mark.parametrize("scene", ["cornell", "teapot", "proc0", "proc1", "proc2", "mix-glass"])
def test_wavefront_pipeline_bit_exact(gpu_tb, settings, scene, sort):
    """SoA-queue wavefront pipeline (generate/extend/shade/connect kernels, ballot-prefix compaction, frames of a
    batch in flight together, ordered accumulation from the sample buffer) against the oracle; the path budget is
    forced small so that several batches and partially filled queues occur.  proc1 / proc2: the SSS interior walk as
    queue entries of their own (glass, mix-free "vol" feature set; rounds until the queues are empty).  sort = 1:
    wf_shade shades every segment in material order (option wavefront_sort) -- same bits."""
    if scene == "cornell":
        gpu_tb.LoadScene(CORNELL); W, H, F, depth = 200, 120, 7, 8
    elif scene == "teapot":
        gpu_tb.LoadScene(TEAPOT); W, H, F, depth = 96, 54, 3, 5
    elif scene == "proc0":
        gpu_tb.LoadProcedural(0, 30000, 11); W, H, F, depth = 120, 72, 4, 6
    elif scene == "proc1":
        gpu_tb.LoadProcedural(1, 30000, 7); W, H, F, depth = 120, 72, 4, 6
    elif scene == "proc2":
        gpu_tb.LoadProcedural(2, 60000, 9); W, H, F, depth = 120, 72, 3, 16
    else:
        gpu_tb.LoadScene(MIX_GLASS); W, H, F, depth = 96, 64, 4, 7
    s = copy.copy(settings); s.MaxBounces = depth
    gpu_tb.SetOption("pipeline", 2); gpu_tb.SetOption("wavefront_paths", W * H * 3); gpu_tb.SetOption("wavefront_sort", sort)
    gpu_tb.SetOption("wavefront_segment", 1024 if scene == "proc2" else 4096)
    try:
        gpu_tb.Render(W, H, F - 2, s, 0.0)
        gpu_tb.Render(W, H, 2, s, 0.0)   # progressive: second call continues the accumulation
        out, jit = gpu_tb.ReadAccumulation(jittered=True)
        variant = gpu_tb.GetOption("last_variant")
        assert gpu_tb.GetOption("last_pipeline") == 2
    finally:
        gpu_tb.SetOption("pipeline", 0); gpu_tb.SetOption("wavefront_paths", 16 << 20); gpu_tb.SetOption("wavefront_sort", 0); gpu_tb.SetOption("wavefront_segment", 4096)
    if scene in ("proc1", "proc2"): assert variant == 5          # "sss": SSS interior walk, no mix materials
    if scene == "mix-glass": assert variant == 3                 # "vol": SSS + mix
    ref = _oracle(gpu_tb, W, H, F, s, jittered=True)
    assert np.array_equal(bits(out), bits(ref["output"]))
    assert np.array_equal(bits(jit), bits(ref["jittered"]))


@pytest.mark.parametrize("refill", [1, 24, 64])
@pytest.mark.parametrize("scene", ["cornell", "proc0", "proc1"])
def test_wavefront_extend_with_refill_bit_exact(gpu_tb, settings, scene, refill):
    """wf_extend in its persistent form (option wavefront_refill = n: a wave claims new queue entries whenever n of its lanes are idle;
    resumable per-lane walks): the same hits, hence the same picture, for a refill at every finished lane (1), at 24 idle lanes and
    only when the whole wave is idle (64)."""
    if scene == "cornell": gpu_tb.LoadScene(CORNELL); W, H, F, depth = 200, 120, 5, 8
    elif scene == "proc0": gpu_tb.LoadProcedural(0, 30000, 11); W, H, F, depth = 120, 72, 4, 6
    else: gpu_tb.LoadProcedural(1, 30000, 7); W, H, F, depth = 120, 72, 4, 6
    s = copy.copy(settings); s.MaxBounces = depth
    gpu_tb.SetOption("pipeline", 2); gpu_tb.SetOption("wavefront_refill", refill); gpu_tb.SetOption("wavefront_paths", W * H * 2)
    try:
        gpu_tb.Render(W, H, F, s, 0.0)
        out = gpu_tb.ReadAccumulation()
        assert gpu_tb.GetOption("last_pipeline") == 2
    finally:
        gpu_tb.SetOption("pipeline", 0); gpu_tb.SetOption("wavefront_refill", 0); gpu_tb.SetOption("wavefront_paths", 16 << 20)
    assert np.array_equal(bits(out), bits(_oracle(gpu_tb, W, H, F, s)["output"]))


@pytest.mark.parametrize("paths", [1, 2])
@pytest.mark.parametrize("scene", ["cornell", "teapot", "proc0"])
def test_pooled_pipeline_bit_exact(gpu_tb, settings, scene, paths):
    """Pipeline 3 (pt_pooled.inc): LDS ray pool with dynamic ray fetch, the bounce ray issued together with the shadow
    feeler, 1 or 2 samples per lane in flight, ordered accumulation from the sample buffer -- against the oracle.
    The sample budget is forced small so that several batches occur; odd sizes leave lanes without a pixel."""
    if scene == "cornell":
        gpu_tb.LoadScene(CORNELL); W, H, F, depth = 200, 120, 7, 8
    elif scene == "teapot":
        gpu_tb.LoadScene(TEAPOT); W, H, F, depth = 97, 55, 3, 5
    else:
        gpu_tb.LoadProcedural(0, 30000, 11); W, H, F, depth = 120, 72, 4, 6
    s = copy.copy(settings); s.MaxBounces = depth
    gpu_tb.SetOption("pipeline", 3); gpu_tb.SetOption("pooled_paths", paths); gpu_tb.SetOption("pooled_samples", W * H * 3)
    try:
        gpu_tb.Render(W, H, F - 2, s, 0.0)
        gpu_tb.Render(W, H, 2, s, 0.0)   # progressive: second call continues the accumulation
        out, jit = gpu_tb.ReadAccumulation(jittered=True)
    finally:
        gpu_tb.SetOption("pipeline", 0); gpu_tb.SetOption("pooled_samples", 256 << 20); gpu_tb.SetOption("pooled_paths", 2)
    ref = _oracle(gpu_tb, W, H, F, s, jittered=True)
    assert np.array_equal(bits(out), bits(ref["output"]))
    assert np.array_equal(bits(jit), bits(ref["jittered"]))


@pytest.mark.parametrize("scene", ["cornell", "teapot", "proc0", "proc2"])
def test_gpu_lbvh_build_equals_host_build(gpu_tb, settings, scene):
    """Row f3: the LBVH built on the GPU (bvh_kernels.hip: Morton codes, rocPRIM radix sort, Karras hierarchy, bottom-up fit)
    is byte-identical to the host builder's layout-A image (itself checked against oracle/bvh_ref.cpp) and gives the same
    tree depth and the same picture."""
    from tracerboy_amd import api
    import ctypes as C

    def load(tb):
        if scene == "cornell": tb.LoadScene(CORNELL)
        elif scene == "teapot": tb.LoadScene(TEAPOT)
        elif scene == "proc0": tb.LoadProcedural(0, 200000, 5)
        else: tb.LoadProcedural(2, 40000, 9)

    def image(tb):
        v = tb.HostSceneView()
        return np.ctypeslib.as_array(C.cast(v.bvh, C.POINTER(C.c_uint8)), shape=(v.bvhBytes,)).copy()

    try:
        gpu_tb.SetOption("bvh_builder", 0); load(gpu_tb)
        host_img, host_depth = image(gpu_tb), gpu_tb.SceneInfo().bvhMaxDepth
        W, H, F = 96, 64, 2
        gpu_tb.Render(W, H, F, settings, 0.0); a = gpu_tb.ReadAccumulation()
        gpu_tb.SetOption("bvh_builder", 2); load(gpu_tb)
        dev_img = image(gpu_tb)
        assert dev_img.shape == host_img.shape and np.array_equal(dev_img, host_img)
        assert gpu_tb.SceneInfo().bvhMaxDepth == host_depth
        gpu_tb.Render(W, H, F, settings, 0.0)
        assert np.array_equal(bits(a), bits(gpu_tb.ReadAccumulation()))
    finally:
        gpu_tb.SetOption("bvh_builder", 0)


@pytest.mark.parametrize("scene", ["cornell", "teapot", "proc0", "proc2", "deep"])
def test_gpu_treelet_build_equals_host_build(gpu_tb, settings, scene):
    """Row f3: LBVH + the fallback layer's three treelet passes (TreeletReorder.hlsl, FindTreelets.hlsl) built on the GPU
    (option bvh_builder = 4: one wave per climbing group) is byte-identical to host builder 3 (itself checked against
    oracle/bvh_ref.cpp), run twice to show that the result does not depend on which group reaches a node first; same
    depth, same picture, and fewer box tests than the plain LBVH."""
    from tracerboy_amd import api
    import ctypes as C

    def load(tb):
        if scene == "cornell": tb.LoadScene(CORNELL)
        elif scene == "teapot": tb.LoadScene(TEAPOT)
        elif scene == "proc0": tb.LoadProcedural(0, 200000, 5)
        elif scene == "proc2": tb.LoadProcedural(2, 40000, 9)
        else: tb.LoadProcedural(1, 700000, 3)   # deep tree: long climbs, many meeting groups

    def image(tb):
        v = tb.HostSceneView()
        return np.ctypeslib.as_array(C.cast(v.bvh, C.POINTER(C.c_uint8)), shape=(v.bvhBytes,)).copy()

    W, H, F = 96, 64, 2
    try:
        gpu_tb.SetOption("bvh_builder", 0); load(gpu_tb)
        plain = image(gpu_tb)
        gpu_tb.SetOption("count_rays", 1); gpu_tb.Render(W, H, 1, settings, 0.0); boxes_plain = gpu_tb.ReadbackStats().rays.boxesTested
        gpu_tb.SetOption("count_rays", 0); gpu_tb.InvalidateHistory()
        gpu_tb.Render(W, H, F, settings, 0.0); a = gpu_tb.ReadAccumulation()
        gpu_tb.SetOption("bvh_builder", 3); load(gpu_tb)
        host_img, host_depth = image(gpu_tb), gpu_tb.SceneInfo().bvhMaxDepth
        assert not np.array_equal(host_img, plain)
        for _ in range(2):
            gpu_tb.SetOption("bvh_builder", 4); load(gpu_tb)
            dev_img = image(gpu_tb)
            assert dev_img.shape == host_img.shape and np.array_equal(dev_img, host_img)
            assert gpu_tb.SceneInfo().bvhMaxDepth == host_depth
        gpu_tb.Render(W, H, F, settings, 0.0)
        assert np.array_equal(bits(a), bits(gpu_tb.ReadAccumulation()))
        gpu_tb.SetOption("count_rays", 1); gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, 1, settings, 0.0)
        assert gpu_tb.ReadbackStats().rays.boxesTested < boxes_plain
    finally:
        gpu_tb.SetOption("bvh_builder", 0); gpu_tb.SetOption("count_rays", 0)


def test_alpha_tested_geometry_bit_exact(gpu_tb, settings):
    """Rows a12 / f1: the IsValidHit alpha test (SharedHitGroup.h:157-179) as a filter on candidate hits of non-opaque
    geometry, on a card textured with a PNG that has transparent texels.  Off (the reference's software path compiles
    it out) and on, both against the oracle; the two pictures differ."""
    import oracle_lib as ol
    scene = os.path.join(GOLDEN, "scenes", "alpha-card", "scene.pbrt")
    gpu_tb.LoadScene(scene)
    W, H, F = 96, 64, 3
    s = copy.copy(settings); s.MaxBounces = 3
    pictures = []
    try:
        for on in (0, 1):
            gpu_tb.SetOption("alpha_test", on); ol.set_alpha_test(on)
            gpu_tb.InvalidateHistory()
            gpu_tb.Render(W, H, F, s, 0.0)
            out = gpu_tb.ReadAccumulation()
            ref = _oracle(gpu_tb, W, H, F, s)
            assert np.array_equal(bits(out), bits(ref["output"])), on
            pictures.append(out)
    finally:
        gpu_tb.SetOption("alpha_test", 0); ol.set_alpha_test(0)
    assert np.any(pictures[0] != pictures[1])


@pytest.mark.parametrize("triangles", [1, 2, 3])
def test_tiny_scenes_every_builder_and_pipeline(gpu_tb, settings, triangles, tmp_path):
    """Edge cases of the tree: a single triangle (the root reference is a leaf, no inner node exists), two (one inner node)
    and three triangles, one of them degenerate (zero area) -- every BVH builder (host LBVH, SAH, GPU LBVH) and every
    pipeline against the oracle."""
    shapes = [
        'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [-1 0 -1  1 0 -1  0 2 -1]',
        'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [-3 0 2  3 0 2  0 0 -4]',
        'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0.5 0.5 0  0.5 0.5 0  0.5 0.5 0]',
    ][:triangles]
    scene = """LookAt 0 1 5  0 1 0  0 1 0
Camera "perspective" "float fov" [45]
Film "image" "integer xresolution" [48] "integer yresolution" [32]
WorldBegin
MakeNamedMaterial "M" "string type" ["matte"] "rgb Kd" [0.6 0.5 0.4]
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [9 9 9]
  %s
AttributeEnd
NamedMaterial "M"
%s
WorldEnd
""" % (shapes[0], "\n".join(shapes[1:]))
    p = tmp_path / "tiny.pbrt"; p.write_text(scene)
    W, H, F = 48, 32, 2
    try:
        for builder in (0, 1, 2, 3, 4):
            gpu_tb.SetOption("bvh_builder", builder)
            gpu_tb.LoadScene(str(p))
            assert gpu_tb.SceneInfo().numTriangles == triangles
            ref = None
            for pipeline in (0, 1, 2, 3):
                gpu_tb.SetOption("pipeline", pipeline); gpu_tb.InvalidateHistory()
                gpu_tb.Render(W, H, F, settings, 0.0)
                out = gpu_tb.ReadAccumulation()
                if ref is None: ref = _oracle(gpu_tb, W, H, F, settings)["output"]
                assert np.array_equal(bits(out), bits(ref)), (builder, pipeline)
    finally:
        gpu_tb.SetOption("pipeline", 0); gpu_tb.SetOption("bvh_builder", 0)


@pytest.mark.parametrize("guided", [0, 2])
@pytest.mark.parametrize("group", [1, 3, 8])
def test_frame_group_mode_bit_exact(gpu_tb, settings, group, guided):
    """Frame-group mode of the persistent kernel (TbDeviceTargets::samples): workgroups render `group` frames each into a
    (frame, pixel) sample buffer that is summed in frame order afterwards -- the image and the jittered image must be the
    oracle's bits, also across two progressive calls and several sample-buffer batches.  guided = 2: the groups of a region shrink
    towards the end of every launch (option guided_groups; by default only calls that wait get them)."""
    gpu_tb.LoadScene(CORNELL)
    W, H, F = 200, 120, 7
    gpu_tb.SetOption("frame_group", group); gpu_tb.SetOption("pooled_samples", W * H * 3); gpu_tb.SetOption("guided_groups", guided)
    try:
        gpu_tb.Render(W, H, F - 2, settings, 0.0)
        gpu_tb.Render(W, H, 2, settings, 0.0)
        out, jit = gpu_tb.ReadAccumulation(jittered=True)
    finally:
        gpu_tb.SetOption("frame_group", 0); gpu_tb.SetOption("pooled_samples", 256 << 20); gpu_tb.SetOption("guided_groups", 1)
    ref = _oracle(gpu_tb, W, H, F, settings, jittered=True)
    assert np.array_equal(bits(out), bits(ref["output"]))
    assert np.array_equal(bits(jit), bits(ref["jittered"]))


def test_frame_group_many_frames_small_image(gpu_tb, settings):
    """Maximum-size edge of the frame-group launch: thousands of frames of a tiny image with frame_group forced to 1 would need
    more frame groups per region than a claimed work item can name (group << 20 | region, 12 bits): the host widens the groups.
    5 000 frames of a 24x20 image (one full and one ragged 16x16 region per row) against the one-pixel-per-lane kernel, bit for bit,
    and the weights count every frame."""
    gpu_tb.LoadScene(CORNELL)
    W, H, F = 24, 20, 5000
    s = copy.copy(settings); s.MaxBounces = 3
    gpu_tb.SetOption("frame_group", 1)
    try:
        gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, F, s, 0.0)
        groups, groups_jit = gpu_tb.ReadAccumulation(jittered=True)
    finally:
        gpu_tb.SetOption("frame_group", -1)
    try:
        gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, F, s, 0.0)
        classic, classic_jit = gpu_tb.ReadAccumulation(jittered=True)
    finally:
        gpu_tb.SetOption("frame_group", 0)
    assert np.all(groups[..., 3] == float(F))
    assert np.array_equal(bits(groups), bits(classic)) and np.array_equal(bits(groups_jit), bits(classic_jit))


@pytest.mark.parametrize("frames,group", [(64, 32), (37, 4), (9, 2), (130, 16), (5, 1)])
def test_guided_frame_groups_bit_exact(gpu_tb, settings, frames, group):
    """Groups that shrink towards the end of a launch (pt_scene.h tb_fg_groups; what a synchronous call of 2+ groups gets by default): slots of
    different sizes in one launch, ragged frame counts, the asynchronous path forced (option 2), a frame whose last region row is half outside --
    every picture the bits of the one-pixel-per-lane kernel, which other tests hold to the oracle."""
    gpu_tb.LoadScene(CORNELL)                      # a scene in LDS: the feature sets' frame-group kernels for such scenes carry the copy that reads a slot's size
    W, H = 328, 200
    s = copy.copy(settings); s.MaxBounces = 5
    gpu_tb.SetOption("frame_group", -1)
    try:
        gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, frames, s, 0.0); ref = gpu_tb.ReadAccumulation(jittered=True)
    finally:
        gpu_tb.SetOption("frame_group", 0)
    gpu_tb.SetOption("frame_group", group)
    try:
        for mode in (1, 2, 0):
            gpu_tb.SetOption("guided_groups", mode)
            gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, frames, s, 0.0)
            assert gpu_tb.GetOption("last_plan_guided_groups") == (1 if (mode and frames >= 2 * group) else 0)
            got = gpu_tb.ReadAccumulation(jittered=True)
            assert np.array_equal(bits(got[0]), bits(ref[0])) and np.array_equal(bits(got[1]), bits(ref[1])), mode
            if mode == 2:                                       # ... and back-to-back asynchronous calls on the two side streams
                for _ in range(3):
                    gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, frames, s, 0.0, sync=False)
                gpu_tb.Sync()
                assert np.array_equal(bits(gpu_tb.ReadAccumulation()), bits(ref[0]))
        # a scene fetched from memory keeps equal groups whatever the option says (its kernels have no such copy), same bits
        gpu_tb.LoadProcedural(0, 20000, 5)
        gpu_tb.SetOption("frame_group", -1); gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, frames, s, 0.0); ref = gpu_tb.ReadAccumulation()
        gpu_tb.SetOption("frame_group", group); gpu_tb.SetOption("guided_groups", 2); gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, frames, s, 0.0)
        assert gpu_tb.GetOption("last_plan_guided_groups") == 0 and np.array_equal(bits(gpu_tb.ReadAccumulation()), bits(ref))
    finally:
        gpu_tb.SetOption("frame_group", 0); gpu_tb.SetOption("guided_groups", 1)


def _read_device_u32(ptr, n):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    buf = np.zeros(n, np.uint32)
    assert hip.hipMemcpy(ctypes.c_void_p(buf.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(buf.nbytes), 2) == 0
    return buf


@pytest.mark.parametrize("world,rank", [(1, 0), (3, 1)])
def test_costly_regions_first_is_a_permutation_and_changes_no_bit(gpu_tb, settings, world, rank):
    """Costly regions first (pt_scene.h TbDeviceTargets::regionOrder, option costly_first): the kernels with interior walks count long walks per region
    and the next launch hands those regions out first.  The table every launch reads must be a permutation of its regions whatever the counts are
    (a region handed out twice or never is a wrong picture), counted regions in front; and the picture is the one-pixel-per-lane kernel's bits with
    the table empty, filled, and filled under back-to-back asynchronous launches that count while the next table is built."""
    gpu_tb.LoadProcedural(1, 20000, 5)               # glass blobs: feature set with interior walks, fetched from memory
    W, H, F = 328, 200, 6                            # last region row half outside the frame
    s = copy.copy(settings); s.MaxBounces = 6
    gpu_tb.SetTileAssignment(rank, world, 64, 64)
    try:
        gpu_tb.SetOption("frame_group", -1); gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, F, s, 0.0); ref = gpu_tb.ReadAccumulation(jittered=True)
        gpu_tb.SetOption("frame_group", 2)
        for rnd in range(3):
            if rnd == 2: gpu_tb.SetOption("costly_late_samples", 256 * 2 * 300)   # only the last 300 items of the usual list count as late
            gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, F, s, 0.0)
            assert gpu_tb.GetOption("last_plan_costly_first") == 1
            got = gpu_tb.ReadAccumulation(jittered=True)
            assert np.array_equal(bits(got[0]), bits(ref[0])) and np.array_equal(bits(got[1]), bits(ref[1])), rnd
            regions, groups = ((W + 15) // 16) * ((H + 15) // 16), F // 2
            cost = _read_device_u32(gpu_tb.GetOption("debug_region_cost_ptr"), 1 << 20)
            if world == 1:
                order = _read_device_u32(gpu_tb.GetOption("debug_region_order_ptr"), 1 + regions * groups)
                usual = [(g << 20) | r for g in range(groups) for r in range(regions)]
                assert sorted(order[1:].tolist()) == usual                              # every item once
                moved = int(order[0])
                assert (rnd == 0) == (moved == 0)                                       # the first launch found no counts; it left some
                pos = {v: i for i, v in enumerate(usual)}
                for part in (order[1:1 + moved], order[1 + moved:]):                    # both parts in their usual order
                    assert np.all(np.diff(np.array([pos[int(v)] for v in part], np.int64)) > 0)
                bx = (W + 15) // 16
                c_of = lambda r: int(cost[(r // bx) << 10 | (r % bx)])
                assert all(pos[int(v)] >= (len(usual) - 300 if rnd == 2 else 0) for v in order[1:1 + moved])
                assert all(c_of(int(v) & 0xfffff) > 0 for v in order[1:1 + moved])      # (a count may have arrived after the table was built: no claim about the rest)
            assert world > 1 or int(cost.sum()) > 0
        for _ in range(4):
            gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, F, s, 0.0, sync=False)
        gpu_tb.Sync()
        assert np.array_equal(bits(gpu_tb.ReadAccumulation()), bits(ref[0]))
        gpu_tb.SetOption("costly_first", 0); gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, F, s, 0.0)
        assert gpu_tb.GetOption("last_plan_costly_first") == 0 and np.array_equal(bits(gpu_tb.ReadAccumulation()), bits(ref[0]))
    finally:
        gpu_tb.SetOption("frame_group", 0); gpu_tb.SetOption("costly_first", 1); gpu_tb.SetOption("costly_late_samples", 1 << 40); gpu_tb.SetTileAssignment(0, 1, 64, 64)


def test_frame_group_default_and_classic_agree(gpu_tb, settings):
    """A call of 8 or more frames takes the frame-group mode by itself (lanes draw (pixel, frame) pairs of their region from an
    LDS counter); frame_group = -1 keeps the one-pixel-per-lane kernel.  Both are the oracle's bits, on a frame whose last
    region row is half outside the image."""
    gpu_tb.LoadScene(CORNELL)
    W, H, F = 200, 120, 12
    gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, F, settings, 0.0)
    auto_out, auto_jit = gpu_tb.ReadAccumulation(jittered=True)
    gpu_tb.SetOption("frame_group", -1)
    try:
        gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, F, settings, 0.0)
        classic_out, classic_jit = gpu_tb.ReadAccumulation(jittered=True)
    finally:
        gpu_tb.SetOption("frame_group", 0)
    ref = _oracle(gpu_tb, W, H, F, settings, jittered=True)
    for out, jit in ((auto_out, auto_jit), (classic_out, classic_jit)):
        assert np.array_equal(bits(out), bits(ref["output"]))
        assert np.array_equal(bits(jit), bits(ref["jittered"]))


def test_async_calls_overlap_and_stay_ordered(gpu_tb, settings):
    """Back-to-back tb_render_async calls: the path-tracing launches alternate between the context's two side streams (each
    may start while the one before drains) but the folds stay in order on the main stream -- progressive accumulation over
    three calls, several sample-buffer batches each, is the oracle's image; with overlap_launches = 0 as well."""
    gpu_tb.LoadScene(CORNELL)
    W, H = 200, 120
    ref = _oracle(gpu_tb, W, H, 8 + 9 + 10, settings, jittered=True)
    for overlap in (1, 0):
        gpu_tb.SetOption("overlap_launches", overlap); gpu_tb.SetOption("pooled_samples", W * H * 4)
        try:
            gpu_tb.InvalidateHistory()
            for n in (8, 9, 10):
                gpu_tb.Render(W, H, n, settings, 0.0, sync=False)
            gpu_tb.Sync()
            out, jit = gpu_tb.ReadAccumulation(jittered=True)
        finally:
            gpu_tb.SetOption("overlap_launches", 1); gpu_tb.SetOption("pooled_samples", 256 << 20)
        assert np.array_equal(bits(out), bits(ref["output"])) and np.array_equal(bits(jit), bits(ref["jittered"]))


def test_tile_split_reproduces_the_single_gpu_image(gpu_tb, settings):
    """Multi-GPU partition on one device: every rank's tiles, packed and un-permuted, give the same bits."""
    from tracerboy_amd import api
    import torch
    gpu_tb.LoadScene(CORNELL)
    W, H, F, world, tw, th = 200, 120, 2, 3, 32, 16
    gpu_tb.SetTileAssignment(0, 1)
    gpu_tb.Render(W, H, F, settings, 0.0)
    full = gpu_tb.ReadAccumulation()
    packed = []
    try:
        for r in range(world):
            gpu_tb.SetTileAssignment(r, world, tw, th)
            gpu_tb.Render(W, H, F, settings, 0.0)
            buf = torch.zeros((gpu_tb.OwnedPixels(W, H), 4), dtype=torch.float32, device="cuda:0")
            torch.cuda.synchronize()  # the library packs on its own stream; order it after torch's fill
            gpu_tb.PackOwnedTo(buf.data_ptr())
            packed.append(buf.cpu().numpy())
    finally:
        gpu_tb.SetTileAssignment(0, 1)
    assert np.array_equal(bits(api.unpack_gathered(W, H, world, tw, th, packed)), bits(full))
    # the device-side un-permute rank 0 runs behind the gather (tb_unpack_gathered_device): same frame, in HBM
    cap = max(p.shape[0] for p in packed)
    gathered = torch.zeros((world, cap, 4), dtype=torch.float32, device="cuda:0")
    for r, p in enumerate(packed): gathered[r, :p.shape[0]] = torch.from_numpy(p).to("cuda:0")
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    gpu_tb.UnpackGatheredTo(gathered.data_ptr(), cap, W, H, world, tw, th, frame.data_ptr()); gpu_tb.Sync()
    assert np.array_equal(bits(frame.cpu().numpy()), bits(full))
    with pytest.raises(api.TracerBoyError):
        gpu_tb.UnpackGatheredTo(gathered.data_ptr(), 16, W, H, world, tw, th, frame.data_ptr())   # capacity too small for rank 0's tiles


def test_full_size_properties_c2(gpu_tb, settings):
    """BASELINE.json configs[1] shape (1920x1080, depth 8) at 2 spp: size-independent properties --
    weights count the frames, no NaN, row strip equals the oracle bit-for-bit, energy is plausible."""
    gpu_tb.LoadScene(CORNELL)
    s = copy.copy(settings); s.MaxBounces = 8
    W, H, F = 1920, 1080, 2
    gpu_tb.Render(W, H, F, s, 0.0)
    out = gpu_tb.ReadAccumulation()
    assert np.all(out[..., 3] == float(F)) and not np.isnan(out).any() and (out[..., :3] >= 0).all()
    ref = ol.render(gpu_tb.HostSceneView(), gpu_tb.FrameConstants(W, H, 0, s, 0.0), W, H, F, y0=536, y1=544, threads=8)["output"]
    assert np.array_equal(bits(out[536:544]), bits(ref[536:544]))
    mean = (out[..., :3] / out[..., 3:]).mean(axis=(0, 1))
    assert 0.05 < mean[0] < 1.0 and mean[0] > mean[1] > mean[2]  # warm light: R > G > B


@pytest.mark.parametrize("scene", ["cornell", "proc", "cornell-bench-tree"])
def test_full_size_frame_groups_equal_one_pixel_per_lane(gpu_tb, settings, scene):
    """The bench workloads in full (1920x1080; cornell 64 spp depth 8 from LDS, 200 k triangles 16 spp depth 6 from global
    memory): the frame-group launch (resident grid, work items claimed through the device counters, sample buffer + ordered
    fold) gives the bits of the one-pixel-per-lane launch over the whole frame, and an 8-row strip of it is the oracle's."""
    W, H = 1920, 1080
    s = copy.copy(settings)
    if scene == "cornell":
        gpu_tb.LoadScene(CORNELL); s.MaxBounces = 8; F = 64
    elif scene == "cornell-bench-tree":           # the headline exactly: bench.WORKLOADS["c2"]'s tree, size, sample count and depth
        w = load_bench_workload(gpu_tb, "c2"); s.MaxBounces = w["depth"]; F = w["spp"]
        assert (W, H) == (w["W"], w["H"])
    else:
        gpu_tb.LoadProcedural(0, 200000, 1234); s.MaxBounces = 6; F = 16
    gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, F, s, 0.0)
    groups, groups_jit = gpu_tb.ReadAccumulation(jittered=True)
    gpu_tb.SetOption("frame_group", -1)
    try:
        gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, F, s, 0.0)
        classic, classic_jit = gpu_tb.ReadAccumulation(jittered=True)
    finally:
        gpu_tb.SetOption("frame_group", 0)
    assert np.all(groups[..., 3] == float(F))
    assert np.array_equal(bits(groups), bits(classic)) and np.array_equal(bits(groups_jit), bits(classic_jit))
    ref = ol.render(gpu_tb.HostSceneView(), gpu_tb.FrameConstants(W, H, 0, s, 0.0), W, H, F, y0=536, y1=544, threads=8)["output"]
    assert np.array_equal(bits(groups[536:544]), bits(ref[536:544]))


def _strip_and_properties(gpu_tb, s, W, H, F, rows):
    """Size-independent checks of a full-size render: every weight counts the frames, nothing is NaN or negative, and the
    8-row strips `rows` are the oracle's bits."""
    out, jit = gpu_tb.ReadAccumulation(jittered=True)
    assert np.all(out[..., 3] == float(F)) and not np.isnan(out).any() and (out[..., :3] >= 0).all()
    assert np.all(jit[..., 3] <= float(F)) and np.all(jit[..., 3] >= 1.0)      # frame 0 always lands in the jittered surface (RayGenCommon.h:715-727)
    view, pf = gpu_tb.HostSceneView(), gpu_tb.FrameConstants(W, H, 0, s, 0.0)
    for y0 in rows:
        ref = ol.render(view, pf, W, H, F, y0=y0, y1=y0 + 8, threads=8, jittered=True)
        assert np.array_equal(bits(out[y0:y0 + 8]), bits(ref["output"][y0:y0 + 8])), y0
        assert np.array_equal(bits(jit[y0:y0 + 8]), bits(ref["jittered"][y0:y0 + 8])), y0
    return out


@pytest.mark.parametrize("tree", ["sah-default-passes", "bench"])
def test_config_c3_dragon_class_870k(gpu_tb, settings, tree):
    """tree = "bench": the tree bench.py's roofline_c3 leg times (bench.WORKLOADS["c3"]: builder and reinsertion options imported).
    BASELINE.json configs[2] shape: the 870 k-triangle procedural stand-in for Scenes/dragon, 1920x1080, depth 6,
    constant white environment.  (a) 2 spp: weights / NaN properties and two 8-row strips against the oracle;
    (b) the configuration's full 128 spp: the frame-group launch (what bench.py times) is bit-identical to the one-pixel-per-lane
    launch over the whole frame, and a strip of the 128-spp image is the oracle's."""
    W, H = 1920, 1080
    s = copy.copy(settings); s.MaxBounces = 6
    if tree == "bench":
        w = load_bench_workload(gpu_tb, "c3")
        assert (w["W"], w["H"], w["spp"], w["depth"]) == (W, H, 128, 6)
    else:
        gpu_tb.SetOption("bvh_builder", 1)                               # binned SAH + the library's own reinsertion passes
        try:
            gpu_tb.LoadProcedural(0, 870000, 1234)
        finally:
            gpu_tb.SetOption("bvh_builder", 0)
    info = gpu_tb.SceneInfo()
    assert abs(info.numTriangles - 870000) <= 8700
    gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, 2, s, 0.0)
    _strip_and_properties(gpu_tb, s, W, H, 2, (400, 536))
    gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, 128, s, 0.0)
    groups, groups_jit = gpu_tb.ReadAccumulation(jittered=True)
    assert np.all(groups[..., 3] == 128.0) and not np.isnan(groups).any()
    gpu_tb.SetOption("frame_group", -1)
    try:
        gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, 128, s, 0.0)
        classic, classic_jit = gpu_tb.ReadAccumulation(jittered=True)
    finally:
        gpu_tb.SetOption("frame_group", 0)
    assert np.array_equal(bits(groups), bits(classic)) and np.array_equal(bits(groups_jit), bits(classic_jit))
    ref = ol.render(gpu_tb.HostSceneView(), gpu_tb.FrameConstants(W, H, 0, s, 0.0), W, H, 128, y0=540, y1=542, threads=8)["output"]
    assert np.array_equal(bits(groups[540:542]), bits(ref[540:542]))


@pytest.mark.parametrize("tree", ["lbvh+treelets-gpu", "bench"])
@pytest.mark.parametrize("cfg", ["c4_van_class", "c5_bistro_class"])
def test_config_c4_c5_4k_scenes(gpu_tb, settings, cfg, tree):
    """tree = "bench": the trees bench.py's roofline_c4 / _c5 and scale_c4 / _c5 legs time (bench.WORKLOADS imported).
    BASELINE.json configs[3] / [4] shapes on ONE GPU (the 8-GPU run splits exactly this frame into tiles): 3840x2160,
    C4-class = 0.7 M triangles with matte / plastic / metal / mirror / glass (SSS walk), default depth 6;
    C5-class = 2.98 M triangles, 40 materials, 4 area lights, depth 16, tree built on the GPU (LBVH + treelet passes).
    2 spp: properties over the whole frame + 8-row strips against the oracle; then the tile split of the 8-rank run
    (this rank = 3 of 8) gives the same bits for its own pixels."""
    W, H, F = 3840, 2160, 2
    s = copy.copy(settings)
    if tree == "bench":
        w = load_bench_workload(gpu_tb, "c4" if cfg == "c4_van_class" else "c5")
        assert (w["W"], w["H"]) == (W, H)
        s.MaxBounces = w["depth"]
    else:
        gpu_tb.SetOption("bvh_builder", 4)
        try:
            if cfg == "c4_van_class": gpu_tb.LoadProcedural(1, 700000, 1234); s.MaxBounces = 6
            else: gpu_tb.LoadProcedural(2, 2980000, 1234); s.MaxBounces = 16
        finally:
            gpu_tb.SetOption("bvh_builder", 0)
    gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, F, s, 0.0)
    assert gpu_tb.GetOption("last_variant") == 5                           # "sss": SSS walk, no mix materials
    full = _strip_and_properties(gpu_tb, s, W, H, F, (1000, 1400))
    try:
        gpu_tb.SetTileAssignment(3, 8)
        gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, F, s, 0.0)
        mine = gpu_tb.ReadAccumulation()
    finally:
        gpu_tb.SetTileAssignment(0, 1)
    owned = np.zeros((H, W), bool)
    tx = (W + 63) // 64
    for t in range(3, tx * ((H + 63) // 64), 8):
        x0, y0 = (t % tx) * 64, (t // tx) * 64
        owned[y0:y0 + 64, x0:x0 + 64] = True
    assert np.array_equal(bits(mine[owned]), bits(full[owned]))


@pytest.mark.parametrize("order", [0, 1, 3, 4, 5])
def test_node_orders_and_banded_items_do_not_change_the_picture(gpu_tb, settings, order):
    """Storage order of the layout-B nodes (option node_order: breadth-first, depth-first, van-Emde-Boas blocks, sibling pairs
    with padding nodes) and the XCD-banded work-item lists of frame-group launches (banded_items) only move data and work around:
    same bits as the default layout, which test_full_size_* compare with the oracle."""
    W, H, F = 200, 120, 9
    s = copy.copy(settings); s.MaxBounces = 5
    gpu_tb.SetOption("bvh_builder", 1)
    try:
        gpu_tb.LoadProcedural(0, 30000, 11)
        gpu_tb.Render(W, H, F, s, 0.0); ref = gpu_tb.ReadAccumulation()
        nodes_default = gpu_tb.SceneInfo().bvhNodesB
        gpu_tb.SetOption("node_order", order); gpu_tb.SetOption("banded_items", 1)
        gpu_tb.LoadProcedural(0, 30000, 11)
        if order in (4, 5): assert gpu_tb.SceneInfo().bvhNodesB >= nodes_default      # padding nodes keep sibling pairs in one 128-B line
        gpu_tb.Render(W, H, F, s, 0.0)
        assert np.array_equal(bits(gpu_tb.ReadAccumulation()), bits(ref))
    finally:
        gpu_tb.SetOption("node_order", 2); gpu_tb.SetOption("banded_items", 0); gpu_tb.SetOption("bvh_builder", 0)
    assert np.array_equal(bits(ref), bits(_oracle_after_reload(gpu_tb, W, H, F, s)))


def _oracle_after_reload(gpu_tb, W, H, F, s):
    gpu_tb.SetOption("bvh_builder", 1)
    try:
        gpu_tb.LoadProcedural(0, 30000, 11)
    finally:
        gpu_tb.SetOption("bvh_builder", 0)
    return _oracle(gpu_tb, W, H, F, s)["output"]


def test_bench_multi_rank_step_on_one_gpu(tmp_path):
    """bench.py --gpus 2 as the driver types it, on a one-GPU box: both ranks share device 0 and the gather goes through host memory
    (TB_BENCH_SHARE_DEVICE / TB_BENCH_BACKEND=gloo; RCCL refuses two ranks on one device) -- everything else is the real N > 1 step:
    self-spawn, own-tiles launches, pack, one gather per render, device-side un-permute.  The line must say n_gpus = 2 and that the
    frame rank 0 assembled equals a single-GPU render of the whole frame, bit for bit."""
    import json, subprocess, sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench as bench_mod
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TB_BENCH_SHARE_DEVICE="1", TB_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--width", "712", "--height", "400", "--spp", "6",
                        "--leg-steps", "2"], capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "tiles2" and out["config"]["tile"] == 64
    assert out["config"]["assembled_frame_equals_single_gpu"] is True
    assert out["value"] > 0 and out["scaling"] == "strong"
    assert out["parity"]["ok"] is True and out["parity"]["bit_equal"] is True      # the assembled frame of the last timed step against the oracle (strips)
    # the per-stage breakdown the driver's SCALE record needs to say where a step's time goes, and how many ranks the collective saw
    sb = out["scale_breakdown"]
    assert out["rccl_ranks"] == 2 and out["collective_backend"] == "gloo"
    assert all(sb[k] > 0 for k in ("render_ms", "pack_ms", "gather_ms", "unpack_ms")) and sb["gather_bytes_per_rank"] > 0
    assert set(sb["mean_over_ranks"]) == {"render_ms", "pack_ms", "gather_ms", "unpack_ms"}
    # the N > 1 line carries the fields of the N = 1 line: a roofline block (rank 0's launch) and the CPU baseline leg (rank 0), plus the
    # slowest rank's scene load
    assert out["roofline"]["avg_launch_ms"] > 0 and out["roofline"]["per_rank"]["owned_pixels_rank0"] > 0
    assert out["config"]["scene_load_s"] > 0
    # ... except the CPU baseline, which the contract times at N = 1 only: the N > 1 record carries a marked copy, never a measurement
    assert out["cpu_baseline"]["measured"] is False
    # the configurations that ARE the 8-GPU configs (BASELINE configs[3] / [4]: van-class, bistro-class and the reference's vw-van at
    # 3840x2160) go through the same step in the same run: value, per-stage breakdown, the assembled-frame check, each leg's own numbers
    for leg in ("c4", "c5", "vwvan"):
        sl = out["scale_" + leg]
        assert sl["n_gpus"] == 2 and sl["value"] > 0 and sl["ms_per_step"] > 0 and sl["steps"] >= 1 and "3840x2160" in sl["workload"]
        assert sl["assembled_frame_equals_single_gpu"] is True, leg
        # the leg is quoted on a 32-spp step (the configurations are 256 / 1024 spp), with the 8-spp short step beside it; the frame rank 0
        # assembled in the last timed step passed the parity gate against the oracle; the deal is the workload's own tile size
        assert "%dspp" % bench_mod.SCALE_SPP in sl["workload"] and sl["at_short_steps"]["spp"] == bench_mod.SCALE_SHORT_SPP
        assert sl["at_short_steps"]["value"] > 0 and sl["at_short_steps"]["ms_per_step"] < sl["ms_per_step"]
        assert sl["parity"]["ok"] is True and sl["parity"]["bit_equal"] is True and sl["parity"]["frames"] == bench_mod.SCALE_SPP
        assert sl["tile"] == bench_mod.WORKLOADS[leg].get("tile", bench_mod.TILE)
        sbl = sl["scale_breakdown"]
        assert all(sbl[k] > 0 for k in ("render_ms", "pack_ms", "gather_ms", "unpack_ms")) and sbl["render_max_over_mean"] >= 1.0
        assert sl["bvh_builder"] == bench_mod.builder_label(bench_mod.WORKLOADS[leg])   # the same tree as the N = 1 line's roofline_<leg>
    assert out["scale_vwvan"]["kernel_variant"] == "vol" and out["scale_c4"]["kernel_variant"] == "sss"


def test_headless_cli_native_rccl_gather_plumbing(tmp_path):
    """tracerboy-hip --ranks N gathers the ranks' packed tiles with librccl (dlopen; ncclGroupStart / Recv / Send / GroupEnd on the
    context's stream) and un-permutes them on the device into rank 0's accumulation surface.  One GPU here: TB_CLI_FORCE_RCCL=1
    runs that sequence with a communicator of one rank (self-gather) -- the EXR must be the plain run's, byte for byte."""
    import subprocess
    from tracerboy_amd import api
    cli = os.path.join(os.path.dirname(api.__file__), "tracerboy-hip")
    outs = []
    for force in ("0", "1"):
        out = str(tmp_path / ("o%s.exr" % force))
        env = dict(os.environ, TB_CLI_FORCE_RCCL=force)
        r = subprocess.run([cli, CORNELL, "--width", "200", "--height", "120", "--spp", "5", "--depth", "4", "--blue-noise", "0", "--out", out],
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1]
    # more ranks than GPUs: the rank without a device fails, the launcher ends the others instead of hanging in the communicator
    import torch
    n = torch.cuda.device_count()
    r = subprocess.run([cli, CORNELL, "--width", "64", "--height", "48", "--spp", "1", "--ranks", str(n + 1), "--out", str(tmp_path / "x.exr")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


@pytest.mark.parametrize("cfg", ["c4_256spp", "c5_1024spp"])
def test_config_c4_c5_full_sample_counts(gpu_tb, settings, cfg):
    """BASELINE.json configs[3] / [4] at their FULL sample counts on one GPU: 3840x2160 x 256 spp (depth 6, 0.7 M triangles with
    glass) and x 1024 spp (depth 16, 2.98 M triangles, 40 materials) -- 2.1 and 8.5 G samples through the frame-group launches in
    batches of the sample-buffer budget.  Size-independent properties over the whole frame (every weight counts the frames, no NaN,
    nothing negative) and one full row of the picture against the oracle at the same sample count, bit for bit."""
    W, H = 3840, 2160
    s = copy.copy(settings)
    gpu_tb.SetOption("bvh_builder", 4)
    try:
        if cfg == "c4_256spp": gpu_tb.LoadProcedural(1, 700000, 1234); s.MaxBounces = 6; F = 256
        else: gpu_tb.LoadProcedural(2, 2980000, 1234); s.MaxBounces = 16; F = 1024
    finally:
        gpu_tb.SetOption("bvh_builder", 0)
    gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, F, s, 0.0)
    assert gpu_tb.GetNumberOfSamplesSinceLastInvalidate() == F
    out = gpu_tb.ReadAccumulation()
    assert np.all(out[..., 3] == float(F)) and not np.isnan(out).any() and (out[..., :3] >= 0).all()
    y = 1203
    ref = ol.render(gpu_tb.HostSceneView(), gpu_tb.FrameConstants(W, H, 0, s, 0.0), W, H, F, y0=y, y1=y + 1, threads=1)["output"]
    assert np.array_equal(bits(out[y]), bits(ref[y]))


@pytest.mark.parametrize("scene", ["cornell", "mix-glass"])
def test_select_pixel_readback_stats(gpu_tb, settings, scene):
    """SelectPixel -> ReadbackStats (TracerBoy.h:362-368): OutputDistanceToFirstHit / OutputMaterial of the selected pixel
    (RayGenCommon.h:632-648, kernel.glsl:1370-1371) against the oracle, over several frames (the last frame that hit wins), for
    pixels on different materials -- incl. a mix material, whose id is reported before the mix coin -- and for a pixel that
    misses (cornell's camera sees nothing but the box: use a pixel of the mix-glass scene that sees the background)."""
    gpu_tb.LoadScene(CORNELL if scene == "cornell" else MIX_GLASS)
    W, H, F = 96, 64, 5
    s = copy.copy(settings); s.MaxBounces = 3
    view = gpu_tb.HostSceneView()
    seen = set()
    try:
        for (x, y) in [(10, 32), (48, 60), (48, 5), (85, 32), (40, 40), (60, 44), (30, 20), (2, 2), (93, 61)]:
            gpu_tb.SelectPixel(x, y); gpu_tb.InvalidateHistory()
            gpu_tb.Render(W, H, F, s, 0.0)
            st = gpu_tb.ReadbackStats()
            pf = gpu_tb.FrameConstants(W, H, 0, s, 0.0)
            assert (pf.SelectedPixelX, pf.SelectedPixelY) == (x, y)
            ref = ol.selected_pixel(view, pf, W, H, F)
            if ref is None:
                assert st.SelectedPixelDistance == 0.0 and st.SelectedMaterialID == 0          # cleared with the history, never written
            else:
                assert np.float32(st.SelectedPixelDistance).view(np.uint32) == np.float32(ref[0]).view(np.uint32), (x, y)
                assert st.SelectedMaterialID == ref[1], (x, y)
                seen.add(ref[1])
            # the picture itself is unchanged by selecting a pixel
        out = gpu_tb.ReadAccumulation()
        gpu_tb.SelectPixel(0xffffffff, 0xffffffff); gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, F, s, 0.0)
        assert np.array_equal(bits(out), bits(gpu_tb.ReadAccumulation()))
    finally:
        gpu_tb.SelectPixel(0xffffffff, 0xffffffff)
    assert len(seen) >= 3, seen


def test_material_edit_and_errors(gpu_tb, settings):
    from tracerboy_amd import api
    gpu_tb.LoadScene(CORNELL)
    assert gpu_tb.IsMaterialIDValid(7) and not gpu_tb.IsMaterialIDValid(8)
    m = gpu_tb.GetMaterial(0)
    m.albedo.x, m.albedo.y, m.albedo.z = 0.1, 0.2, 0.9
    gpu_tb.SetMaterial(0, m)
    W, H = 48, 32
    gpu_tb.Render(W, H, 2, settings, 0.0)
    assert np.array_equal(bits(gpu_tb.ReadAccumulation()), bits(_oracle(gpu_tb, W, H, 2, settings)["output"]))
    with pytest.raises(api.TracerBoyError):
        gpu_tb.GetMaterial(99)
    with pytest.raises(api.TracerBoyError):
        gpu_tb.LoadScene("/nonexistent/scene.pbrt")
    gpu_tb.LoadScene(CORNELL)


def test_headless_cli_writes_the_library_result(gpu_tb, settings, tmp_path):
    """tracerboy-hip (cli.cpp, C ABI only): renders cornell-box with the GPU-built treelet tree and writes linear radiance as
    OpenEXR and the tonemapped back buffer as PNG; the EXR pixels are the library's accumulation divided by its weight."""
    import struct, subprocess
    from tracerboy_amd import api
    cli = os.path.join(os.path.dirname(api.__file__), "tracerboy-hip")
    W, H, F = 64, 48, 9
    exr, png = str(tmp_path / "o.exr"), str(tmp_path / "o.png")
    for out in (exr, png):
        r = subprocess.run([cli, CORNELL, "--width", str(W), "--height", str(H), "--spp", str(F), "--depth", "4", "--blue-noise", "0",
                            "--builder", "treelets-gpu", "--out", out], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
    assert open(png, "rb").read(8) == b"\x89PNG\r\n\x1a\n"
    b = open(exr, "rb").read()
    assert struct.unpack_from("<I", b, 0)[0] == 20000630
    start = len(b) - H * (8 + 16 * W)                     # H scan-line blocks of {y, bytes, A B G R planes} close the file
    px = np.stack([np.frombuffer(b, np.float32, 4 * W, start + y * (8 + 16 * W) + 8).reshape(4, W) for y in range(H)])   # [y][A B G R][x]
    s = copy.copy(settings); s.MaxBounces = 4
    gpu_tb.SetOption("bvh_builder", 4)
    try:
        gpu_tb.LoadScene(CORNELL); gpu_tb.Render(W, H, F, s, 0.0)
        acc = gpu_tb.ReadAccumulation()
    finally:
        gpu_tb.SetOption("bvh_builder", 0)
    inv = np.where(acc[..., 3] > 0, np.float32(1.0) / acc[..., 3], np.float32(0.0)).astype(np.float32)
    for c, plane in ((0, 3), (1, 2), (2, 1)):
        assert np.array_equal(bits(px[:, plane, :]), bits(acc[..., c] * inv))
    assert np.array_equal(px[:, 0, :], (acc[..., 3] > 0).astype(np.float32))


@pytest.mark.parametrize("size", [(1, 1), (7, 5), (13, 9), (65, 33), (257, 1), (1, 129), (72, 40)])
def test_ragged_frame_sizes_bit_exact(gpu_tb, settings, size):
    """Frames that are not multiples of the 8 x 8 thread group (the reference's dispatch rounds up and its shader returns for pixels
    outside, SoftwareRayTraceCS.hlsl:33-37), down to a single pixel and single rows / columns: the persistent megakernel with and
    without frame groups and the wavefront pipeline against the oracle, the jittered surface and the sample weights included."""
    W, H = size
    gpu_tb.LoadScene(CORNELL)
    for frames in (1, 9):
        ref = _oracle(gpu_tb, W, H, frames, settings, jittered=True)
        for pipeline, groups in ((0, 0), (0, -1), (0, 4), (2, 0)):      # frame_group 0: by itself from 8 frames on; -1: one pixel per lane; 4: groups of four
            try:
                gpu_tb.SetOption("pipeline", pipeline); gpu_tb.SetOption("frame_group", groups)
                gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, frames, settings, 0.0)
                out, jit = gpu_tb.ReadAccumulation(jittered=True)
            finally:
                gpu_tb.SetOption("pipeline", 0); gpu_tb.SetOption("frame_group", 0)
            assert out.shape == (H, W, 4)
            assert np.array_equal(bits(out), bits(ref["output"])), (size, frames, pipeline, groups)
            assert np.array_equal(bits(jit), bits(ref["jittered"])), (size, frames, pipeline, groups)


def test_empty_and_degenerate_inputs_are_refused_or_rendered(gpu_tb, settings, tmp_path):
    """A world without a triangle is refused at load (the reference builds nothing to trace and asserts); zero frames is a no-op;
    a mesh of only zero-area triangles and a non-finite vertex still render what the oracle renders (no hang, no fault)."""
    from tracerboy_amd import api
    head = 'LookAt 0 1 5  0 1 0  0 1 0\nCamera "perspective" "float fov" [45]\nWorldBegin\n'
    p = tmp_path / "empty.pbrt"; p.write_text(head + "WorldEnd\n")
    with pytest.raises(api.TracerBoyError):
        gpu_tb.LoadScene(str(p))
    gpu_tb.LoadScene(CORNELL)
    gpu_tb.InvalidateHistory(); gpu_tb.Render(32, 16, 0, settings, 0.0)
    assert gpu_tb.GetNumberOfSamplesSinceLastInvalidate() == 0
    gpu_tb.Render(32, 16, 2, settings, 0.0)
    assert np.array_equal(bits(gpu_tb.ReadAccumulation()), bits(_oracle(gpu_tb, 32, 16, 2, settings)["output"]))
    flat = head + 'Shape "trianglemesh" "integer indices" [0 1 2 0 2 1] "point P" [1 1 0  1 1 0  1 1 0]\nWorldEnd\n'
    p = tmp_path / "flat.pbrt"; p.write_text(flat)
    gpu_tb.LoadScene(str(p)); gpu_tb.Render(40, 24, 2, settings, 0.0)
    assert np.array_equal(bits(gpu_tb.ReadAccumulation()), bits(_oracle(gpu_tb, 40, 24, 2, settings)["output"]))
    nonfinite = head + ('MakeNamedMaterial "M" "string type" ["matte"] "rgb Kd" [0.6 0.5 0.4]\nNamedMaterial "M"\n'
                        'Shape "trianglemesh" "integer indices" [0 1 2 3 4 5] "point P" [-3 0 2  3 0 2  0 0 -4   0 1 0  1e39 1 0  0 2 0]\nWorldEnd\n')
    p = tmp_path / "nonfinite.pbrt"; p.write_text(nonfinite)
    gpu_tb.LoadScene(str(p)); gpu_tb.Render(40, 24, 2, settings, 0.0)
    out = gpu_tb.ReadAccumulation(); ref = _oracle(gpu_tb, 40, 24, 2, settings)["output"]
    assert np.array_equal(bits(out), bits(ref))
    gpu_tb.LoadScene(CORNELL)


def test_c_abi_misuse_returns_error_codes(gpu_tb, settings):
    """The boundary never aborts (the reference asserts; the C ABI returns a negative code and keeps a message): null
    handles and buffers, calls in the wrong order, unknown options, absurd sizes -- through raw ctypes, not the Python wrapper."""
    import ctypes as C
    from tracerboy_amd import api
    L = gpu_tb._L
    null = C.c_void_p(None)
    assert L.tb_render(null, 8, 8, 1, None, C.c_float(0)) < 0
    assert L.tb_read_accum(null, None, None) < 0
    assert L.tb_load_scene(null, b"x.pbrt") < 0
    assert L.tb_set_option(null, b"aov", 1) < 0
    assert L.tb_group_size(null) == 0
    L.tb_destroy(null); L.tb_invalidate_history(null)                   # no-ops
    fresh = api.TracerBoy()                                              # a second context on the same device, no scene yet
    try:
        h = fresh._ctx
        assert L.tb_render(h, 8, 8, 1, None, C.c_float(0)) < 0 and b"no scene" in L.tb_last_error(h)
        buf = (C.c_float * (8 * 8 * 4))()
        assert L.tb_read_accum(h, buf, None) < 0
        assert L.tb_get_camera(h, None) < 0
        assert L.tb_load_scene(h, None) < 0
        assert L.tb_load_scene(h, b"/nonexistent/dir/scene.pbrt") < 0 and len(L.tb_last_error(h)) > 0
        fresh.LoadScene(CORNELL)
        assert L.tb_render(h, 0, 8, 1, None, C.c_float(0)) < 0
        assert L.tb_render(h, 1 << 20, 1 << 20, 1, None, C.c_float(0)) < 0      # refused, not attempted
        assert L.tb_read_aov(h, 3, buf) < 0                                       # AOVs were not enabled
        assert L.tb_read_aov(h, 99, buf) < 0
        assert L.tb_get_material(h, -1, None) < 0 and L.tb_set_material(h, 10 ** 6, None) < 0
        assert L.tb_set_tile_assignment(h, 3, 2, 64, 64) < 0                     # rank beyond the world
        assert L.tb_set_tile_assignment(h, 0, 1, 0, 64) < 0                      # zero-sized tiles
        # the context is still usable after all of that
        fresh.Render(24, 16, 2, settings, 0.0)
        gpu_tb.LoadScene(CORNELL); gpu_tb.InvalidateHistory(); gpu_tb.Render(24, 16, 2, settings, 0.0)
        assert np.array_equal(bits(fresh.ReadAccumulation()), bits(gpu_tb.ReadAccumulation()))
    finally:
        fresh.close()
