#!/usr/bin/env python3
"""Stress of the frame groups that shrink over the end of a launch (round 6; the GUIDED copy of the LDS-resident frame-group kernels): cornell-box,
small frames whose buffers stay L2-resident, three random streams over the two alternating sample buffers (a slot nobody rendered, or rendered twice
into the wrong frame, shows as a mismatch with the oracle), group sizes 1 / 2 / 4 / 8, frame counts with ragged ends, synchronous calls (the default
policy) and forced on back-to-back asynchronous calls.   python scripts/guided_stress.py [reps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
from tracerboy_amd import api
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
tb = api.TracerBoy()
tb.LoadScene(os.path.join(ROOT, "tests", "golden", "scenes", "cornell-box", "scene.pbrt"))
s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 6
def bits(a): return np.ascontiguousarray(a).view(np.uint32)
total = bad = 0
for (W, H, F, G) in ((200, 120, 9, 2), (200, 120, 37, 4), (328, 200, 19, 1), (72, 40, 64, 8), (200, 120, 5, 2)):
    refs = [ol.render(tb.HostSceneView(), tb.FrameConstants(W, H, 0, s, float(t)), W, H, F, threads=8)["output"] for t in (0, 1, 2)]
    tb.SetOption("frame_group", G)
    for mode in (1, 2):
        tb.SetOption("guided_groups", mode)
        b0 = bad
        for rep in range(reps):
            t = rep % 3
            tb.InvalidateHistory()
            if mode == 2:      # two asynchronous calls back to back on the two side streams, the second one's picture is checked
                tb.Render(W, H, F, s, float((t + 1) % 3), sync=False); tb.InvalidateHistory(); tb.Render(W, H, F, s, float(t), sync=False); tb.Sync()
            else:
                tb.Render(W, H, F, s, float(t))
            assert tb.GetOption("last_plan_guided_groups") == (1 if F >= 2 * G else 0)
            total += 1
            if not np.array_equal(bits(tb.ReadAccumulation()), bits(refs[t])): bad += 1
        print("%dx%dx%d frames, groups of %d, guided_groups=%d (%s): %d renders, %d bad" % (W, H, F, G, mode, "waits" if mode == 1 else "async pairs", reps, bad - b0), flush=True)
tb.SetOption("frame_group", 0); tb.SetOption("guided_groups", 1)
print("total %d renders, %d bad" % (total, bad))
sys.exit(1 if bad else 0)
