#!/usr/bin/env python3
"""How often does a hit record of the primary-visibility pre-pass fail its validation (launch epoch + check word) and get walked again?
Round 3 attributed such records to per-XCD L2s keeping a line across a kernel boundary; scripts/microbench/l2_stale.hip finds that
boundary coherent in every launch shape of the product.  This counts what the build itself sees: renders of three scenes x three random
streams over the two alternating record buffers, pre-pass forced.   python scripts/prepass_reject_count.py [renders per size] [out.json]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tracerboy_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
tb = api.TracerBoy(0); tb.SetOption("primary_prepass", 2)
s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 6
rows = []
for (W, H, F, reps) in ((200, 120, 9, n), (640, 360, 8, n // 3), (1920, 1080, 8, n // 15)):
    for kind, tris, seed in ((1, 30000, 7), (0, 30000, 5)):
        tb.LoadProcedural(kind, tris, seed)
        before = tb.GetOption("debug_prepass_rejects"); used = 0
        for rep in range(reps):
            tb.InvalidateHistory(); tb.Render(W, H, F, s, float(rep % 3), sync=(rep % 5 != 4)); used += tb.GetOption("last_primary_prepass")
        tb.Sync()
        row = {"frame": "%dx%dx%d" % (W, H, F), "scene": "proc%d:%d" % (kind, tris), "renders": reps, "with_prepass": used, "records": reps * W * H * F, "rejected": tb.GetOption("debug_prepass_rejects") - before}
        rows.append(row); print(json.dumps(row), flush=True)
if len(sys.argv) > 2: json.dump(rows, open(sys.argv[2], "w"), indent=1)
