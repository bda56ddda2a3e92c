#!/usr/bin/env python3
"""Teapot (the reference's own scene: 126 k triangles, environment-lit, feature set `surf`) with and without the primary-visibility pre-pass."""
import copy, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracerboy_amd import api
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tb = api.TracerBoy()
s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 8
tb.SetOption("bvh_builder", 1); tb.LoadScene(os.path.join(root, "tests/golden/scenes/Teapot/scene.pbrt")); tb.SetOption("bvh_builder", 0)
W, H, F = 1920, 1080, 16
res = {}; row = {"scene": "Teapot 1920x1080x16 depth 8", "lights": None}
for pre in (0, 2, 1):
    tb.SetOption("primary_prepass", pre); ts = []
    for r in range(5):
        tb.InvalidateHistory(); t = time.perf_counter(); tb.Render(W, H, F, s, 0.0); ts.append(time.perf_counter() - t)
    res[pre] = tb.ReadAccumulation()
    row[{0: "never", 2: "asked for", 1: "default"}[pre]] = {"Msamples_per_s": round(W * H * F / np.median(ts[1:]) / 1e6, 1), "used": bool(tb.GetOption("last_primary_prepass")), "variant": int(tb.GetOption("last_variant"))}
row["bit_identical"] = bool(np.array_equal(res[0].view(np.uint32), res[2].view(np.uint32)) and np.array_equal(res[0].view(np.uint32), res[1].view(np.uint32)))
print(json.dumps(row))
if len(sys.argv) > 1: json.dump(row, open(sys.argv[1], "w"), indent=1)
