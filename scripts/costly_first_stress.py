#!/usr/bin/env python3
"""Stress of costly-regions-first (round 6; pt_scene.h TbDeviceTargets::regionOrder): a glass scene fetched from memory, small frames, every launch's item
list rebuilt from counts that the launch before is still adding to -- back-to-back asynchronous pairs on the two side streams, whole frame and ranks
of a tile split, several group sizes, three random streams, a different list every pair (option costly_late_samples).  A region handed out twice or never shows as a mismatch with the one-pixel-per-lane
kernel's picture (which tests/ hold to the oracle).   python scripts/costly_first_stress.py [reps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tracerboy_amd import api
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
tb = api.TracerBoy()
s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 6
def bits(a): return np.ascontiguousarray(a).view(np.uint32)
total = bad = 0
for scene in ((1, 20000, 5), (2, 30000, 7)):
    tb.LoadProcedural(*scene)
    for (W, H, F, G, world, rank) in ((200, 120, 8, 2, 1, 0), (328, 200, 6, 1, 1, 0), (328, 200, 12, 4, 3, 1), (520, 296, 4, 2, 8, 5), (72, 40, 32, 8, 1, 0)):
        tb.SetTileAssignment(rank, world, 64, 64)
        tb.SetOption("frame_group", -1)
        refs = []
        for t in (0, 1, 2):
            tb.InvalidateHistory(); tb.Render(W, H, F, s, float(t)); refs.append(tb.ReadAccumulation().copy())
        tb.SetOption("frame_group", G)
        b0 = bad
        for rep in range(reps):
            t = rep % 3
            # a different list every pair (how much of the usual list counts as late): the table a side stream's launch reads is not the one its last launch read
            tb.SetOption("costly_late_samples", (1 << 40, 256 * G * 37, 256 * G * 5)[(rep // 3) % 3])
            tb.InvalidateHistory(); tb.Render(W, H, F, s, float((t + 1) % 3), sync=False); tb.InvalidateHistory(); tb.Render(W, H, F, s, float(t), sync=False); tb.Sync()
            assert tb.GetOption("last_plan_costly_first") == 1
            total += 1
            if not np.array_equal(bits(tb.ReadAccumulation()), bits(refs[t])): bad += 1
        print("proc%d:%d %dx%dx%d frames, groups of %d, rank %d of %d: %d async pairs, %d bad" % (scene[0], scene[1], W, H, F, G, rank, world, reps, bad - b0), flush=True)
tb.SetOption("frame_group", 0); tb.SetOption("costly_late_samples", 1 << 40); tb.SetTileAssignment(0, 1, 64, 64)
print("total %d renders, %d bad" % (total, bad))
sys.exit(1 if bad else 0)
