#!/usr/bin/env python3
"""Frame-group size on the LDS-resident cornell-box, launches enqueued back to back like bench.py's timed steps (1920x1080x64, depth 8, SAH)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracerboy_amd import api
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tb = api.TracerBoy()
s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 8
tb.SetOption("bvh_builder", 1); tb.LoadScene(os.path.join(root, "tests/golden/scenes/cornell-box/scene.pbrt")); tb.SetOption("bvh_builder", 0)
W, H, F = 1920, 1080, 64
for g in (0, 4, 8, 16, 32, 64):
    tb.SetOption("frame_group", g)
    for _ in range(3): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0)
    best = 0
    for rep in range(3):
        tb.Sync(); t = time.perf_counter()
        for _ in range(10): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0, sync=False)
        tb.Sync(); dt = time.perf_counter() - t
        best = max(best, W * H * F * 10 / dt / 1e6)
    print("cornell-box frame_group", g if g else "auto", "%.1f Msamples/s" % best, flush=True)
