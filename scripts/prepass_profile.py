#!/usr/bin/env python3
"""One scene, `reps` renders with the primary-visibility pre-pass on (argv[1] = 1) or off (0): run under rocprofv3 --kernel-trace --stats
to see the pre-pass and the lock-step kernel side by side.  argv[2]: c3 | c4 | c5"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracerboy_amd import api
pre = int(sys.argv[1]); which = sys.argv[2] if len(sys.argv) > 2 else "c3"
proc, W, H, F, depth = {"c3": ((0, 870000, 1234), 1920, 1080, 32, 6), "c4": ((1, 700000, 1234), 3840, 2160, 8, 6), "c5": ((2, 2980000, 1234), 3840, 2160, 8, 16)}[which]
tb = api.TracerBoy()
s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = depth
tb.SetOption("bvh_builder", 4); tb.LoadProcedural(*proc)
tb.SetOption("primary_prepass", 2 if pre else 0)
if len(sys.argv) > 3: tb.SetOption("stack_lds_cap", int(sys.argv[3])); tb.SetOption("stack_overflow_max", 64)
for r in range(4):
    tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0)
print("done", tb.GetOption("last_primary_prepass"))
