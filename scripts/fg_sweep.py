#!/usr/bin/env python3
"""Frame-group size with the primary-visibility pre-pass asked for: 870 k scene 1080p x 32 and van-class 4K x 8."""
import copy, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracerboy_amd import api
tb = api.TracerBoy()
s0 = api.GetDefaultOutputSettings(); s0.EnableBlueNoise = 0
for name, proc, W, H, F, depth in (("c3", (0, 870000, 1234), 1920, 1080, 32, 6), ("c4", (1, 700000, 1234), 3840, 2160, 8, 6)):
    s = copy.copy(s0); s.MaxBounces = depth
    tb.SetOption("bvh_builder", 4); tb.LoadProcedural(*proc); tb.SetOption("bvh_builder", 0); tb.SetOption("primary_prepass", 2)
    for g in (0, 1, 2, 4, 8, 16, 32):
        if g > F: continue
        tb.SetOption("frame_group", g); ts = []
        for r in range(4):
            tb.InvalidateHistory(); t = time.perf_counter(); tb.Render(W, H, F, s, 0.0); ts.append(time.perf_counter() - t)
        print(name, "frame_group", g if g else "auto", "%.1f Msamples/s" % (W * H * F / np.median(ts[1:]) / 1e6), flush=True)
    tb.SetOption("frame_group", 0)
