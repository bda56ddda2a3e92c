#!/usr/bin/env python3
"""Two-level (instanced) traversal against the flattened scene: K x K instances of one displaced-sphere object
of T triangles each over a ground quad, under an area light; the same .pbrt loaded with flatten_instances = 1 (one BVH over
K*K*T triangles, the tuned single-level walk) and = 0 (top level over instances + one bottom-level structure, the two-level walk).
Prints Msamples/s, box / triangle tests per sample and the BVH memory of both.

  python scripts/instanced_bench.py [--grid 8] [--tris 8192] [--spp 16] [--builder 4]   (>= 500 k instanced triangles by default)
"""
import argparse, json, math, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tracerboy_amd import api  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=8); ap.add_argument("--tris", type=int, default=8192); ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080); ap.add_argument("--builder", type=int, default=3)
ap.add_argument("--material", default="matte")
a = ap.parse_args()

rings = max(4, int(math.sqrt(a.tris / 2))); segs = max(4, a.tris // (2 * rings))
P, N, I = [], [], []
for r in range(rings + 1):
    for s in range(segs + 1):
        th = math.pi * r / rings; ph = 2 * math.pi * (s % segs) / segs
        rad = 0.42 * (1 + 0.12 * math.sin(5 * th) * math.sin(4 * ph) + 0.05 * math.sin(9 * ph + 2 * th))
        d = (math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph))
        P += ["%.6f %.6f %.6f" % (d[0] * rad, d[1] * rad, d[2] * rad)]; N += ["%.5f %.5f %.5f" % d]
for r in range(rings):
    for s in range(segs):
        p0 = r * (segs + 1) + s; p1 = p0 + 1; p2 = p0 + segs + 1; p3 = p2 + 1
        if r != 0: I += ["%d %d %d" % (p0, p1, p3)]
        if r != rings - 1: I += ["%d %d %d" % (p0, p3, p2)]
tris_per_object = len(I)
K = a.grid; ext = K * 1.0
mat = {"matte": '"string type" ["matte"] "rgb Kd" [0.6 0.45 0.3]', "glass": '"string type" ["glass"] "float index" [1.5]', "plastic": '"string type" ["plastic"] "rgb Kd" [0.2 0.4 0.6] "rgb Ks" [0.3 0.3 0.3] "float roughness" [0.15]'}[a.material]
lines = ['LookAt 0 %.3f %.3f  0 0.3 0  0 1 0' % (0.75 * ext, 1.25 * ext), 'Camera "perspective" "float fov" [38]', 'Film "image" "integer xresolution" [%d] "integer yresolution" [%d]' % (a.width, a.height), "WorldBegin",
         'MakeNamedMaterial "Ground" "string type" ["matte"] "rgb Kd" [0.5 0.5 0.5]', 'MakeNamedMaterial "Blob" ' + mat,
         'NamedMaterial "Ground"', "AttributeBegin", '  AreaLightSource "diffuse" "rgb L" [9 9 8]',
         '  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [%g %g %g  %g %g %g  %g %g %g  %g %g %g] "normal N" [0 -1 0 0 -1 0 0 -1 0 0 -1 0]' % (-ext / 3, ext, -ext / 3, ext / 3, ext, -ext / 3, ext / 3, ext, ext / 3, -ext / 3, ext, ext / 3),
         "AttributeEnd", 'NamedMaterial "Ground"',
         'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [%g 0 %g  %g 0 %g  %g 0 %g  %g 0 %g] "normal N" [0 1 0 0 1 0 0 1 0 0 1 0]' % (-ext, ext, ext, ext, ext, -ext, -ext, -ext),
         'NamedMaterial "Blob"', 'ObjectBegin "blob"', '  Shape "trianglemesh" "integer indices" [%s] "point P" [%s] "normal N" [%s]' % (" ".join(I), " ".join(P), " ".join(N)), "ObjectEnd"]
for i in range(K):
    for j in range(K):
        x = (i + 0.5) / K * 2 * ext - ext; z = (j + 0.5) / K * 2 * ext - ext
        lines += ["AttributeBegin", "  Translate %.4f %.4f %.4f" % (x * 0.9, 0.5 + 0.1 * ((i * 7 + j * 3) % 5), z * 0.9), "  Rotate %d 0 1 0" % ((i * 37 + j * 91) % 360), "  Scale %.3f %.3f %.3f" % (0.8 + 0.05 * ((i + j) % 5), 0.9, 0.8 + 0.04 * ((i * j) % 6)), '  ObjectInstance "blob"', "AttributeEnd"]
lines += ["WorldEnd"]
path = os.path.join(tempfile.mkdtemp(), "instanced.pbrt"); open(path, "w").write("\n".join(lines) + "\n")

s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 6
res = {"instances": K * K, "triangles_per_object": tris_per_object, "instanced_triangles": K * K * tris_per_object, "frame": "%dx%dx%d" % (a.width, a.height, a.spp), "material": a.material, "bvh_builder": a.builder}
tb = api.TracerBoy(0)
tb.SetOption("bvh_builder", a.builder)
for flat in (1, 0):
    tb.SetOption("flatten_instances", flat)
    t = time.time(); tb.LoadScene(path); load = time.time() - t
    info = tb.SceneInfo()
    tb.Render(a.width, a.height, a.spp, s, 0.0)                     # warm
    ms = []
    for _ in range(3):
        tb.InvalidateHistory(); t = time.perf_counter(); tb.Render(a.width, a.height, a.spp, s, 0.0); ms.append((time.perf_counter() - t) * 1e3)
    variant = ["matte", "env", "surf", "vol", "full", "sss"][tb.GetOption("last_variant")]
    pre = {}
    for opt in (0, 2):           # the primary-visibility pre-pass, never / whenever the kernels have it (flattened scenes only)
        tb.SetOption("primary_prepass", opt); t2 = []
        for _ in range(3):
            tb.InvalidateHistory(); t = time.perf_counter(); tb.Render(a.width, a.height, a.spp, s, 0.0); t2.append((time.perf_counter() - t) * 1e3)
        pre["off" if opt == 0 else "forced"] = {"Msamples_per_s": round(a.width * a.height * a.spp / min(t2) / 1e3, 1), "used": bool(tb.GetOption("last_primary_prepass"))}
    tb.SetOption("primary_prepass", 1)
    tb.SetOption("count_rays", 1); tb.Render(a.width, a.height, 1, s, 0.0); st = tb.ReadbackStats().rays; tb.SetOption("count_rays", 0)
    res["flattened" if flat else "two_level"] = {"Msamples_per_s": round(a.width * a.height * a.spp / min(ms) / 1e3, 1), "ms": round(min(ms), 3), "primary_prepass": pre, "kernel_variant": variant,
                                                 "boxes_per_sample": round(st.boxesTested / st.samples, 2), "tris_per_sample": round(st.trianglesTested / st.samples, 2), "rays_per_sample": round(st.rays / st.samples, 3),
                                                 "triangles_in_bvh": int(info.numTriangles), "bvh_bytes_layout_a": int(info.bvhBytesA), "bvh_depth": int(info.bvhMaxDepth), "load_s": round(load, 2)}
tb.SetOption("flatten_instances", 1)
print(json.dumps(res))
