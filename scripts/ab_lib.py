#!/usr/bin/env python3
"""Times the three big scenes with the library in TB_LIB (or the tree's own): median of 5 synchronous renders, pre-pass off where the option exists."""
import copy, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracerboy_amd import api
tb = api.TracerBoy()
try: tb.SetOption("primary_prepass", 0)
except api.TracerBoyError: pass
s0 = api.GetDefaultOutputSettings(); s0.EnableBlueNoise = 0
for name, proc, W, H, F, depth in (("c3", (0, 870000, 1234), 1920, 1080, 32, 6), ("c4", (1, 700000, 1234), 3840, 2160, 8, 6), ("c5", (2, 2980000, 1234), 3840, 2160, 8, 16)):
    s = copy.copy(s0); s.MaxBounces = depth
    tb.SetOption("bvh_builder", 4); tb.LoadProcedural(*proc); tb.SetOption("bvh_builder", 0)
    ts = []
    for r in range(6):
        tb.InvalidateHistory(); t = time.perf_counter(); tb.Render(W, H, F, s, 0.0); ts.append(time.perf_counter() - t)
    print(name, "Msamples/s %.1f" % (W * H * F / np.median(ts[1:]) / 1e6), "kernel ms %.3f" % (tb.GetOption("last_kernel_us") / 1e3), flush=True)
