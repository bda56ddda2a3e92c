#!/usr/bin/env python3
"""Calibration: the streaming kernel (pipeline 1, resumable walk) against the lock-step kernel in its one-pixel-per-lane form (frame_group -1), same launches."""
import copy, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracerboy_amd import api
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tb = api.TracerBoy()
s0 = api.GetDefaultOutputSettings(); s0.EnableBlueNoise = 0
for name, loader, W, H, F, depth in (("cornell (sah)", lambda: (tb.SetOption("bvh_builder", 1), tb.LoadScene(os.path.join(root, "tests/golden/scenes/cornell-box/scene.pbrt"))), 1920, 1080, 16, 8),
                                     ("870k", lambda: (tb.SetOption("bvh_builder", 4), tb.LoadProcedural(0, 870000, 1234)), 1920, 1080, 16, 6)):
    s = copy.copy(s0); s.MaxBounces = depth
    loader(); tb.SetOption("bvh_builder", 0)
    for label, opts in (("lock-step, frame groups", {"pipeline": 0, "frame_group": 0}), ("lock-step, one pixel per lane", {"pipeline": 0, "frame_group": -1}), ("streaming (pipeline 1)", {"pipeline": 1, "frame_group": -1})):
        for k, v in opts.items(): tb.SetOption(k, v)
        ts = []
        for r in range(4):
            tb.InvalidateHistory(); t = time.perf_counter(); tb.Render(W, H, F, s, 0.0); ts.append(time.perf_counter() - t)
        print("%-16s %-32s %8.1f Msamples/s  (pipeline ran: %d, variant %d)" % (name, label, W * H * F / np.median(ts[1:]) / 1e6, tb.GetOption("last_pipeline"), tb.GetOption("last_variant")), flush=True)
    tb.SetOption("pipeline", 0); tb.SetOption("frame_group", 0)
