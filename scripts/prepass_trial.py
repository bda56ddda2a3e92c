#!/usr/bin/env python3
"""The default pre-pass policy on scenes it has to try: six calls of each scene, what each call did and how long it took."""
import copy, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracerboy_amd import api
tb = api.TracerBoy()
s0 = api.GetDefaultOutputSettings(); s0.EnableBlueNoise = 0
rows = []
for name, proc, W, H, F, depth in (("c4-class", (1, 700000, 1234), 3840, 2160, 8, 6), ("c5-class", (2, 2980000, 1234), 3840, 2160, 8, 16), ("c3", (0, 870000, 1234), 1920, 1080, 32, 6)):
    s = copy.copy(s0); s.MaxBounces = depth
    tb.SetOption("bvh_builder", 4); tb.LoadProcedural(*proc); tb.SetOption("bvh_builder", 0); tb.SetOption("primary_prepass", 1)
    calls = []
    for r in range(7):
        tb.InvalidateHistory(); t = time.perf_counter(); tb.Render(W, H, F, s, 0.0); ms = (time.perf_counter() - t) * 1e3
        calls.append((int(tb.GetOption("last_primary_prepass")), round(ms, 2)))
    ref = tb.ReadAccumulation()
    tb.SetOption("primary_prepass", 0); tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0)
    same = bool(np.array_equal(ref.view(np.uint32), tb.ReadAccumulation().view(np.uint32)))
    row = {"scene": name, "calls (pre-pass used, ms)": calls, "bit_identical_to_never": same}
    print(json.dumps(row), flush=True); rows.append(row)
if len(sys.argv) > 1: json.dump(rows, open(sys.argv[1], "w"), indent=1)
