"""Row f3, second half: two-level (instanced) traversal -- TraverseFunction.hlsli:603-640, the !FAST_PATH branch -- with
flatten_instances = 0.  Fixture tests/golden/scenes/instances: one two-geometry object instanced three times (translation,
rotation + non-uniform scale, mirror), a nested object holding two more instances of it, instanced twice, and a single-triangle
object (its bottom-level root is a leaf), beside world-level shapes (the reference's "global BLAS" under an identity instance).

CPU: the loader against the REFERENCE parser's dump; the structure the host builds (instances, hit-group bases, top-level boxes,
inverse transforms); the oracle's two-level picture against its own picture of the flattened scene.
GPU (-m gpu): closest hits (t, barycentrics, primitive, hit-group index, counters) and radiance bit-exact against the oracle."""
import copy
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol
from conftest import GOLDEN
from tracerboy_amd import _ctypes_abi as abi

SCENE = os.path.join(GOLDEN, "scenes", "instances", "scene.pbrt")
W, H, F = 96, 64, 3


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_loader_matches_reference_parser_on_instances(built, tmp_path):
    """ObjectBegin / ObjectEnd / ObjectInstance incl. nesting; the reference parser does not restore the graphics state at
    ObjectEnd (Parser.inl:621-635) and neither does the build's."""
    from tracerboy_amd import api
    out = str(tmp_path / "dump.txt"); err = C.create_string_buffer(256)
    assert api.lib().tb_host_pbrt_dump(SCENE.encode(), out.encode(), err, 256) == 0, err.value
    assert open(out).read() == open(os.path.join(GOLDEN, "instances.parser.txt")).read()


def _aabb_nodes(img):
    n = (np.frombuffer(img, np.uint32, 1, 4)[0] - 16) // 32
    raw = np.frombuffer(img, np.uint8, n * 32, 16).reshape(n, 32)
    return raw[:, :12].copy().view(np.float32), raw[:, 12:16].copy().view(np.uint32)[:, 0], raw[:, 16:28].copy().view(np.float32), raw[:, 28:32].copy().view(np.uint32)[:, 0]


def test_two_level_structure(built):
    from tracerboy_amd import api
    hs = api.HostScene(SCENE, flatten_instances=False)
    flat = api.HostScene(SCENE, flatten_instances=True)
    v, i = hs.view(), hs.info()
    # world shapes (3) -> structure 0 under an identity instance; gadget x 3 + pair x 2 x 2 nested + shard = 8 more instances
    assert (v.numInstances, v.numBlas) == (9, 3)
    assert i.numTriangles == 6 + 14 + 1 and flat.info().numTriangles == 6 + 7 * 14 + 1
    assert i.numGeometries == 3 + 7 * 2 + 1 == flat.info().numGeometries   # one hit-group record per (instance, geometry)
    assert i.numLights == flat.info().numLights == 2
    offs = [v.blasOffsets[k] for k in range(v.numBlas + 1)]
    assert offs[0] == 0 and offs[-1] == v.bvhBytes and all(o % 16 == 0 for o in offs[:-1])
    tl = C.string_at(v.tlas, v.tlasBytes)
    M = v.numInstances
    off_meta = np.frombuffer(tl, np.uint32, 4, 0)
    assert off_meta[0] == 16 and off_meta[1] == 16 + 32 * (2 * M - 1) and off_meta[3] == v.tlasBytes == off_meta[1] + 116 * M
    c, fx, h, fy = _aabb_nodes(tl)
    md = [abi.TbBvhMetadata.from_buffer_copy(tl, int(off_meta[1]) + 116 * k) for k in range(M)]
    assert sorted(m.InstanceIndex for m in md) == list(range(M))
    bases = sorted(m.InstanceContributionToHitGroupIndexAndFlags & 0xFFFFFF for m in md)
    assert bases == [0, 3, 5, 7, 9, 11, 13, 15, 17]                          # structure 0 has 3 geometries, the gadget 2, the shard 1
    bvh = C.string_at(v.bvh, v.bvhBytes)
    for k, m in enumerate(md):
        w2o = np.array(m.WorldToObject[:], np.float64).reshape(3, 4); o2w = np.array(m.ObjectToWorld[:], np.float64).reshape(3, 4)
        prod = np.vstack([o2w, [0, 0, 0, 1]]) @ np.vstack([w2o, [0, 0, 0, 1]])
        assert np.allclose(prod, np.eye(4), atol=2e-6)                         # InverseAffineTransform (RayTracingHelper.hlsli:297-316)
        # leaf box = the eight transformed corners of the structure's root box (TransformAABB :318-344)
        bc, _, bh, _ = _aabb_nodes(bvh[offs[m.BlasIndex]:offs[m.BlasIndex + 1]])
        lo, hi = (bc[0] - bh[0]).astype(np.float64), (bc[0] + bh[0]).astype(np.float64)
        corners = np.array([[x, y, z, 1.0] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
        wc = corners @ o2w.T
        leaf = M - 1 + k
        assert fx[leaf] == (0x80000000 | k) and fy[leaf] == 1
        assert np.allclose(c[leaf] - h[leaf], wc.min(0), atol=1e-5) and np.allclose(c[leaf] + h[leaf], wc.max(0), atol=1e-5)
    for n in range(M - 1):                                                     # every inner box is the union of its children's
        l, r = fx[n] & 0xFFFFFF, fy[n]
        lo = np.minimum(c[l] - h[l], c[r] - h[r]); hi = np.maximum(c[l] + h[l], c[r] + h[r])
        assert np.allclose(c[n] - h[n], lo, atol=1e-5) and np.allclose(c[n] + h[n], hi, atol=1e-5)
    tri = hs.triangles()
    assert tri["tri_geometry"].max() == 2                                      # geometry index INSIDE a structure
    # the whole top-level image equals the oracle's serial restatement of the fallback layer's top-level build, byte for byte
    by_instance = sorted(md, key=lambda m: m.InstanceIndex)
    roots = []
    for m in by_instance:
        bc, _, bh, _ = _aabb_nodes(bvh[offs[m.BlasIndex]:offs[m.BlasIndex + 1]])
        roots.append(np.concatenate([bc[0] - bh[0], bc[0] + bh[0]]))
    ref = ol.build_tlas(np.array([m.ObjectToWorld[:] for m in by_instance], np.float32), np.array(roots, np.float32),
                        [m.BlasIndex for m in by_instance], [m.InstanceContributionToHitGroupIndexAndFlags & 0xFFFFFF for m in by_instance])
    assert ref.tobytes() == tl


def test_many_instances_top_level_equals_oracle_build(built, tmp_path):
    """300 instances with random rotations, non-uniform scales and translations (duplicate Morton codes included): the host's
    top-level image is the oracle restatement's bytes, and two-level closest hits agree with the flattened scene's."""
    from tracerboy_amd import api
    rng = np.random.default_rng(17)
    lines = ['LookAt 0 6 30  0 0 0  0 1 0', 'Camera "perspective" "float fov" [40]', 'Film "image" "integer xresolution" [64] "integer yresolution" [48]', 'WorldBegin',
             'MakeNamedMaterial "A" "string type" ["matte"] "rgb Kd" [0.6 0.5 0.4]', 'NamedMaterial "A"', 'ObjectBegin "tet"',
             'Shape "trianglemesh" "integer indices" [0 1 2 0 1 3 1 2 3 0 2 3] "point P" [0 0 0  1 0 0  0.5 0 0.9  0.5 0.8 0.3]', 'ObjectEnd',
             'AttributeBegin', 'AreaLightSource "diffuse" "rgb L" [5 5 5]',
             'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-3 12 -3 3 12 -3 3 12 3 -3 12 3]', 'AttributeEnd']
    for k in range(300):
        t = rng.uniform(-8, 8, 3) if k % 7 else np.array([1.0, 2.0, 3.0])          # every seventh instance sits at the same place
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        sc = rng.uniform(0.4, 2.0, 3)
        lines += ['AttributeBegin', 'Translate %.6f %.6f %.6f' % tuple(t), 'Rotate %.4f %.6f %.6f %.6f' % ((rng.uniform(0, 360),) + tuple(ax)),
                  'Scale %.6f %.6f %.6f' % tuple(sc), 'ObjectInstance "tet"', 'AttributeEnd']
    lines.append('WorldEnd')
    path = tmp_path / "many.pbrt"; path.write_text("\n".join(lines))
    two = api.HostScene(str(path), flatten_instances=False); flat = api.HostScene(str(path), flatten_instances=True)
    v = two.view()
    assert v.numInstances == 301 and v.numBlas == 2
    tl = C.string_at(v.tlas, v.tlasBytes); bvh = C.string_at(v.bvh, v.bvhBytes)
    offs = [v.blasOffsets[k] for k in range(v.numBlas + 1)]
    off_meta = int(np.frombuffer(tl, np.uint32, 1, 4)[0])
    md = sorted((abi.TbBvhMetadata.from_buffer_copy(tl, off_meta + 116 * k) for k in range(v.numInstances)), key=lambda m: m.InstanceIndex)
    roots = []
    for m in md:
        bc, _, bh, _ = _aabb_nodes(bvh[offs[m.BlasIndex]:offs[m.BlasIndex + 1]])
        roots.append(np.concatenate([bc[0] - bh[0], bc[0] + bh[0]]))
    ref = ol.build_tlas(np.array([m.ObjectToWorld[:] for m in md], np.float32), np.array(roots, np.float32), [m.BlasIndex for m in md],
                        [m.InstanceContributionToHitGroupIndexAndFlags & 0xFFFFFF for m in md])
    assert ref.tobytes() == tl
    o = rng.uniform(-12, 12, (5000, 3)).astype(np.float32)
    d = (rng.uniform(-8, 8, (5000, 3)) - o).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    a, b = ol.trace_closest(v, o, d), ol.trace_closest(flat.view(), o, d)
    both = (a["t"] > 0) & (b["t"] > 0)
    assert both.sum() > 1000 and (a["t"] > 0).sum() - both.sum() < 5 and (b["t"] > 0).sum() - both.sum() < 5   # grazing rays may differ by rounding
    close = np.isclose(a["t"][both], b["t"][both], rtol=1e-4)
    assert close.mean() > 0.995


def test_two_level_oracle_agrees_with_flattened_oracle(built, settings):
    """The same rays find the same surfaces: closest-hit t of the two-level walk equals the flattened scene's up to the rounding
    of the object-space arithmetic, and the pictures differ only where unrotated instance normals matter (the reference keeps
    instanced vertex attributes in object space, TracerBoy.cpp:1623-1624)."""
    from tracerboy_amd import api
    two = api.HostScene(SCENE, flatten_instances=False); flat = api.HostScene(SCENE, flatten_instances=True)
    rng = np.random.default_rng(5)
    o = np.tile(np.array([[0.5, 3.2, 8.5]], np.float32), (4000, 1))
    d = (rng.uniform([-3.5, 0.0, -2.5], [3.5, 1.6, 4.0], (4000, 3)) - o).astype(np.float32)      # aimed at the objects
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a, b = ol.trace_closest(two.view(), o, d), ol.trace_closest(flat.view(), o, d)
    hit = b["t"] > 0
    assert hit.sum() > 3000 and np.array_equal(a["t"] > 0, hit)
    assert np.allclose(a["t"][hit], b["t"][hit], rtol=2e-5)
    assert np.array_equal(a["material"][hit], b["material"][hit])
    s = copy.copy(settings)
    ia = ol.render(two.view(), two.frame_constants(s, 0, 0.0), W, H, 4, threads=8)["output"]
    ib = ol.render(flat.view(), flat.frame_constants(s, 0, 0.0), W, H, 4, threads=8)["output"]
    assert not np.isnan(ia).any()
    assert abs(ia[..., :3].sum() - ib[..., :3].sum()) / ib[..., :3].sum() < 0.03


@pytest.mark.gpu
def test_two_level_closest_hits_bit_exact(gpu_tb, settings):
    gpu_tb.SetOption("flatten_instances", 0)
    try:
        gpu_tb.LoadScene(SCENE)
    finally:
        gpu_tb.SetOption("flatten_instances", 1)
    view = gpu_tb.HostSceneView()
    assert view.numInstances == 9
    rng = np.random.default_rng(9)
    info = gpu_tb.SceneInfo()
    lo, hi = np.array(info.sceneMin[:]), np.array(info.sceneMax[:])
    ro = rng.uniform(lo - 0.5, hi + 0.5, (6000, 3)).astype(np.float32)
    rd = rng.normal(size=(6000, 3)).astype(np.float32); rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    axis = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 300)] * rng.choice([-1, 1], 300)[:, None]
    O = np.concatenate([ro, ro[:300]]); D = np.concatenate([rd, axis.astype(np.float32)])
    g = gpu_tb.TraceClosest(O, D); c = ol.trace_closest(view, O, D)
    assert (c["t"] > 0).sum() > 2000
    for k in ("t", "bary", "normal", "uv"):
        assert np.array_equal(bits(g[k]), bits(c[k])), k
    for k in ("material", "prim", "geom", "boxes", "tris"):
        assert np.array_equal(g[k], c[k]), k
    assert len(np.unique(c["geom"][c["t"] > 0])) >= 12                       # rays reach most (instance, geometry) records


@pytest.mark.gpu
@pytest.mark.parametrize("builder", [0, 1, 3, 4])
def test_two_level_radiance_bit_exact(gpu_tb, settings, builder):
    gpu_tb.SetOption("flatten_instances", 0); gpu_tb.SetOption("bvh_builder", builder)
    try:
        gpu_tb.LoadScene(SCENE)
    finally:
        gpu_tb.SetOption("flatten_instances", 1); gpu_tb.SetOption("bvh_builder", 0)
    s = copy.copy(settings); s.MaxBounces = 5
    gpu_tb.Render(W, H, F, s, 0.0)
    assert gpu_tb.GetOption("last_variant") == 4                              # the two-level walk lives in the full-feature kernels
    out, jit = gpu_tb.ReadAccumulation(jittered=True)
    ref = ol.render(gpu_tb.HostSceneView(), gpu_tb.FrameConstants(W, H, 0, s, 0.0), W, H, F, threads=8, jittered=True)
    assert not np.isnan(out).any() and out[..., :3].max() > 0
    assert np.array_equal(bits(out), bits(ref["output"])) and np.array_equal(bits(jit), bits(ref["jittered"]))
    # frame-group launches (12 frames) and the counters of the counting kernel agree with the oracle as well
    gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, 12, s, 0.0)
    ref12 = ol.render(gpu_tb.HostSceneView(), gpu_tb.FrameConstants(W, H, 0, s, 0.0), W, H, 12, threads=8)
    assert np.array_equal(bits(gpu_tb.ReadAccumulation()), bits(ref12["output"]))
    gpu_tb.SetOption("count_rays", 1)
    try:
        gpu_tb.Render(W, H, 2, s, 0.0)
        st = gpu_tb.ReadbackStats().rays
        rs = ol.render(gpu_tb.HostSceneView(), gpu_tb.FrameConstants(W, H, 0, s, 0.0), W, H, 2, threads=8, stats=True)["stats"]
        for k in ("boxesTested", "trianglesTested", "hitsShaded", "samples", "rays"):
            assert getattr(st, k) == getattr(rs, k), k
    finally:
        gpu_tb.SetOption("count_rays", 0)


@pytest.mark.gpu
def test_two_level_alpha_tested_instances_bit_exact(gpu_tb, settings):
    """IsValidHit on candidate hits of instanced geometry (RayGenCommon.h:423-434): fixture alpha-card/instanced.pbrt, the alpha-tested
    card as an object instanced twice, walked two-level.  Filter off and on against the oracle -- radiance and closest hits -- and the
    two pictures differ (round 2 skipped the filter in the two-level walk)."""
    scene = os.path.join(GOLDEN, "scenes", "alpha-card", "instanced.pbrt")
    gpu_tb.SetOption("flatten_instances", 0)
    try:
        gpu_tb.LoadScene(scene)
    finally:
        gpu_tb.SetOption("flatten_instances", 1)
    view = gpu_tb.HostSceneView()
    assert view.numInstances == 3
    s = copy.copy(settings); s.MaxBounces = 3
    rng = np.random.default_rng(2)
    o = np.tile(np.array([[0.0, 1.0, 4.0]], np.float32), (5000, 1))
    d = (rng.uniform([-2.0, 0.0, 0.0], [2.0, 2.4, 1.0], (5000, 3)) - o).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pictures, hits = [], []
    try:
        for on in (0, 1):
            gpu_tb.SetOption("alpha_test", on); ol.set_alpha_test(on)
            gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, F, s, 0.0)
            out = gpu_tb.ReadAccumulation()
            ref = ol.render(view, gpu_tb.FrameConstants(W, H, 0, s, 0.0), W, H, F, threads=8)["output"]
            assert np.array_equal(bits(out), bits(ref)), on
            g = gpu_tb.TraceClosest(o, d); c = ol.trace_closest(view, o, d)
            for k in ("t", "bary"): assert np.array_equal(bits(g[k]), bits(c[k])), (on, k)
            for k in ("prim", "geom", "material"): assert np.array_equal(g[k], c[k]), (on, k)
            pictures.append(out); hits.append(g["geom"].copy())
    finally:
        gpu_tb.SetOption("alpha_test", 0); ol.set_alpha_test(0)
    assert np.any(pictures[0] != pictures[1]) and np.any(hits[0] != hits[1])    # rays pass through the transparent texels of the cards


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["matte", "sss"])
def test_two_level_tuned_kernel_copies_bit_exact(gpu_tb, settings, kind, tmp_path):
    """The tuned two-level walk (while-while parking, split stack) of the frame-group kernels in the higher-occupancy copies of the
    matte / env / sss / vol feature sets (TWOLEVEL kernels, pt_variant.inc): the instances fixture with its plastic / metal materials
    turned into matte (feature set `matte`) or glass (feature set `sss`, interior walks through instanced geometry), 9 frames so that
    the launch is a frame-group one, with the stack in LDS and split (stack_lds_cap = 3), against the oracle; a one-frame call of
    the same scene goes through the full-feature kernels and must give the same first frame."""
    text = open(SCENE).read()
    if kind == "matte":
        text = text.replace('"string type" ["plastic"] "rgb Kd" [0.1 0.3 0.7] "rgb Ks" [0.4 0.4 0.4] "float roughness" [0.1]', '"string type" ["matte"] "rgb Kd" [0.1 0.3 0.7]')
        text = text.replace('"string type" ["metal"] "float uroughness" [0.15] "float vroughness" [0.15] "rgb eta" [0.9 0.9 0.9]', '"string type" ["matte"] "rgb Kd" [0.6 0.6 0.65]')
        variant = 0
    else:
        text = text.replace('"string type" ["plastic"] "rgb Kd" [0.1 0.3 0.7] "rgb Ks" [0.4 0.4 0.4] "float roughness" [0.1]', '"string type" ["glass"] "float index" [1.45]')
        variant = 5
    assert text != open(SCENE).read()
    path = tmp_path / "scene.pbrt"; path.write_text(text)
    gpu_tb.SetOption("flatten_instances", 0)
    try:
        gpu_tb.LoadScene(str(path))
    finally:
        gpu_tb.SetOption("flatten_instances", 1)
    view = gpu_tb.HostSceneView()
    assert view.numInstances == 9
    s = copy.copy(settings); s.MaxBounces = 5
    ref = ol.render(view, gpu_tb.FrameConstants(W, H, 0, s, 0.0), W, H, 9, threads=8, jittered=True)
    try:
        for cap in (0, 3):
            gpu_tb.SetOption("stack_lds_cap", cap); gpu_tb.SetOption("stack_overflow_max", 64 if cap else 24)
            gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, 9, s, 0.0)
            assert gpu_tb.GetOption("last_variant") == variant, (kind, cap)          # not the full feature set: the tuned copy ran
            out, jit = gpu_tb.ReadAccumulation(jittered=True)
            assert np.array_equal(bits(out), bits(ref["output"])) and np.array_equal(bits(jit), bits(ref["jittered"])), (kind, cap)
    finally:
        gpu_tb.SetOption("stack_lds_cap", 0); gpu_tb.SetOption("stack_overflow_max", 24)
    gpu_tb.InvalidateHistory(); gpu_tb.Render(W, H, 1, s, 0.0)
    assert gpu_tb.GetOption("last_variant") == 4
    one = ol.render(view, gpu_tb.FrameConstants(W, H, 0, s, 0.0), W, H, 1, threads=8)["output"]
    assert np.array_equal(bits(gpu_tb.ReadAccumulation()), bits(one))
    # the other pipelines have no two-level walk: refused, not silently wrong
    gpu_tb.SetOption("pipeline", 2)
    try:
        with pytest.raises(Exception):
            gpu_tb.Render(W, H, 2, s, 0.0)
    finally:
        gpu_tb.SetOption("pipeline", 0)


@pytest.mark.gpu
@pytest.mark.parametrize("gpu_builder,host_builder", [(2, 0), (4, 3)])
def test_gpu_built_two_level_structures_equal_the_host_build(gpu_tb, gpu_builder, host_builder, tmp_path):
    """bvh_builder 2 / 4 on an instanced scene: every bottom-level structure AND the top level (TopLevelLoadAABBs.hlsli:62-105 leaf
    boxes and metadata, Morton codes of the box centres, sort, Karras, bottom-up fit -- GpuBVH2Builder.cpp:498-501) are constructed on
    the GPU; the images are, byte for byte, the host builders' (which tests above pin to the oracle's serial restatement).  Also on a
    scene with 301 instances (coincident ones included) and on one whose top level is a single instance."""
    from tracerboy_amd import api
    many = ['LookAt 0 6 14  0 0.5 0  0 1 0', 'Camera "perspective" "float fov" [40]', 'Film "image" "integer xresolution" [64] "integer yresolution" [48]', "WorldBegin",
            'MakeNamedMaterial "M" "string type" ["matte"] "rgb Kd" [0.5 0.5 0.5]', 'NamedMaterial "M"',
            'ObjectBegin "t"', '  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3 0 3 1 1 3 2] "point P" [0 0 0  0.4 0 0  0.2 0 0.35  0.2 0.4 0.12]', "ObjectEnd"]
    for i in range(300):
        many += ["AttributeBegin", "  Translate %g %g %g" % ((i * 37 % 17) - 8.0, (i % 5) * 0.3, (i * 11 % 13) - 6.0), "  Rotate %d 0 1 0" % (i * 23 % 360), '  ObjectInstance "t"', "AttributeEnd"]
    many += ["WorldEnd"]
    p_many = tmp_path / "many.pbrt"; p_many.write_text("\n".join(many) + "\n")
    p_one = tmp_path / "one.pbrt"; p_one.write_text("\n".join(many[:9] + ["AttributeBegin", "  Translate 1 0 0", '  ObjectInstance "t"', "AttributeEnd", "WorldEnd"]) + "\n")
    for path, instances in ((SCENE, 9), (str(p_many), 300), (str(p_one), 1)):
        host = api.HostScene(path, bvh_builder=host_builder, flatten_instances=False)
        hv = host.view()
        gpu_tb.SetOption("flatten_instances", 0); gpu_tb.SetOption("bvh_builder", gpu_builder)
        try:
            gpu_tb.LoadScene(path)
        finally:
            gpu_tb.SetOption("flatten_instances", 1); gpu_tb.SetOption("bvh_builder", 0)
        gv = gpu_tb.HostSceneView()
        assert gv.numInstances == hv.numInstances == instances
        assert C.string_at(gv.tlas, gv.tlasBytes) == C.string_at(hv.tlas, hv.tlasBytes), path
        assert C.string_at(gv.bvh, gv.bvhBytes) == C.string_at(hv.bvh, hv.bvhBytes), path
        assert [gv.blasOffsets[k] for k in range(gv.numBlas + 1)] == [hv.blasOffsets[k] for k in range(hv.numBlas + 1)]
        assert gpu_tb.SceneInfo().bvhMaxDepth == host.info().bvhMaxDepth
        o = np.tile(np.array([[0.5, 3.2, 8.5]], np.float32), (2000, 1))
        rng = np.random.default_rng(4); d = rng.normal(size=(2000, 3)).astype(np.float32); d[:, 1] -= 0.5; d[:, 2] -= 1.0; d /= np.linalg.norm(d, axis=1, keepdims=True)
        g = gpu_tb.TraceClosest(o, d); c = ol.trace_closest(gv, o, d)
        assert np.array_equal(bits(g["t"]), bits(c["t"])) and np.array_equal(g["geom"], c["geom"])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["matte", "glass"])
def test_instanced_scene_as_the_first_render_of_a_fresh_context(settings, kind, tmp_path):
    """the first frame-group render of a two-level scene in a FRESH context used to fail -- the stream warm-up launched
    the one-pixel-per-lane form of the tuned copy, which that copy does not have for instanced scenes -- and every later call with it.
    The session-wide context of the other tests had always warmed the launcher with another scene first."""
    from tracerboy_amd import api
    text = open(SCENE).read()
    if kind == "matte":
        text = text.replace('"string type" ["plastic"] "rgb Kd" [0.1 0.3 0.7] "rgb Ks" [0.4 0.4 0.4] "float roughness" [0.1]', '"string type" ["matte"] "rgb Kd" [0.1 0.3 0.7]')
        text = text.replace('"string type" ["metal"] "float uroughness" [0.15] "float vroughness" [0.15] "rgb eta" [0.9 0.9 0.9]', '"string type" ["matte"] "rgb Kd" [0.6 0.6 0.65]')
    else:
        text = text.replace('"string type" ["plastic"] "rgb Kd" [0.1 0.3 0.7] "rgb Ks" [0.4 0.4 0.4] "float roughness" [0.1]', '"string type" ["glass"] "float index" [1.45]')
    p = tmp_path / "scene.pbrt"; p.write_text(text)
    for f in os.listdir(os.path.dirname(SCENE)):
        if f != "scene.pbrt":
            src = os.path.join(os.path.dirname(SCENE), f)
            if os.path.isfile(src): (tmp_path / f).write_bytes(open(src, "rb").read())
    s = copy.copy(settings); s.MaxBounces = 5
    with api.TracerBoy(0) as tb:
        tb.SetOption("flatten_instances", 0)
        tb.LoadScene(str(p))
        tb.Render(W, H, 9, s, 0.0)                                   # 9 frames: a frame-group launch, overlapped, the context's very first render
        out = tb.ReadAccumulation()
        ref = ol.render(tb.HostSceneView(), tb.FrameConstants(W, H, 0, s, 0.0), W, H, 9, threads=8)
        assert np.array_equal(bits(out), bits(ref["output"]))
        tb.InvalidateHistory(); tb.Render(W, H, 9, s, 0.0)           # and again
        assert np.array_equal(bits(tb.ReadAccumulation()), bits(ref["output"]))
