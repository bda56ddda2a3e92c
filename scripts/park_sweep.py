#!/usr/bin/env python3
"""park_min sweep (while-while scheduling of traverse(): leave the inner-node loop when fewer lanes than this still descend) with the
primary-visibility pre-pass on and off: configs[2]-class scene, 1920x1080x32, depth 6."""
import copy, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracerboy_amd import api
tb = api.TracerBoy()
s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 6
W, H, F = 1920, 1080, 32
rows = []
for park in (8, 16, 24, 32, 40, 48):
    tb.SetOption("park_min", park); tb.SetOption("bvh_builder", 4); tb.LoadProcedural(0, 870000, 1234)
    row = {"park_min": park}
    for pre in (0, 1):
        tb.SetOption("primary_prepass", pre); ts = []
        for r in range(4):
            tb.InvalidateHistory(); t = time.perf_counter(); tb.Render(W, H, F, s, 0.0); ts.append(time.perf_counter() - t)
        row["prepass%d_Msamples_s" % pre] = round(W * H * F / np.median(ts[1:]) / 1e6, 1)
    print(json.dumps(row), flush=True); rows.append(row)
if len(sys.argv) > 1: json.dump(rows, open(sys.argv[1], "w"), indent=1)
