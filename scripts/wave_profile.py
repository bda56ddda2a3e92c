#!/usr/bin/env python3
"""Prints the per-phase wave occupancy of the persistent kernel (counting variant) for a workload.
   python scripts/wave_profile.py [--scene cornell-box|procK:N] [--width W --height H --spp S --depth D --builder B]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tracerboy_amd import api  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cornell-box")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=4)
ap.add_argument("--depth", type=int, default=8)
ap.add_argument("--builder", type=int, default=1)
ap.add_argument("--pipeline", type=int, default=0)
ap.add_argument("--paths", type=int, default=2)
a = ap.parse_args()
s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = a.depth
tb = api.TracerBoy(0)
tb.SetOption("bvh_builder", a.builder)
tb.SetOption("pipeline", a.pipeline)
if a.scene == "cornell-box":
    tb.LoadScene(os.path.join(ROOT, "tests/golden/scenes/cornell-box/scene.pbrt"))
elif a.scene.startswith("proc"):
    k, n = a.scene[4:].split(":"); tb.LoadProcedural(int(k), int(n), 1234)
else:
    tb.LoadScene(a.scene)
ap2 = None
if a.pipeline == 3:
    tb.SetOption("pooled_profile", 1); tb.SetOption("pooled_paths", a.paths)
    tb.Render(a.width, a.height, a.spp, s, 0.0)
    prof = tb.WaveProfile()
    n = a.width * a.height * a.spp
    print(json.dumps({"samples": n, "ms": tb.LastRenderMs(), "phases": {k: {"active": v[0], "trips": v[1], "occupancy": round(v[2], 4), "trips_per_64_samples": round(v[1] * 64 / n, 2)} for k, v in prof.items()}}, indent=1))
    sys.exit(0)
tb.SetOption("count_rays", 1)
tb.Render(a.width, a.height, a.spp, s, 0.0)
st = tb.ReadbackStats().rays
prof = tb.WaveProfile()
out = {"samples": st.samples, "rays_per_sample": st.rays / st.samples, "boxes_per_ray": st.boxesTested / st.rays, "tris_per_ray": st.trianglesTested / st.rays,
       "ms": tb.LastRenderMs(), "phases": {k: {"active": v[0], "trips": v[1], "occupancy": round(v[2], 4)} for k, v in prof.items()}}
print(json.dumps(out, indent=1))
