#!/usr/bin/env python3
"""First-bounce pass (option first_bounce, pt_first) against the primary-visibility pre-pass on bench.py's workloads, one process, alternating:
bit equality of the accumulation surfaces and ms per step of asynchronous steps.   python scripts/first_bounce_ab.py [out.json] [legs...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import bench
from tracerboy_amd import api
out_path = sys.argv[1] if len(sys.argv) > 1 else None
legs = sys.argv[2:] or ["c4", "c5", "vwvan", "teapot", "c3"]
b = bench.Bench(api, 0); tb = b.tb
res = {}
for key in legs:
    w = bench.WORKLOADS[key]; W, H, SPP = w["W"], w["H"], (32 if key == "c3" else w["spp"]); s = b.settings(w["depth"])
    b.load_workload(key); tb.SetOption("overlap_launches", 2)
    row = {}
    pics = {}
    for rep in range(2):
        for fb in (0, 1):
            tb.SetOption("first_bounce", fb)
            for _ in range(3): tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0, sync=False)
            tb.Sync(); t = time.perf_counter()
            for _ in range(6): tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0, sync=False)
            tb.Sync(); ms = (time.perf_counter() - t) / 6 * 1e3
            row.setdefault("first_bounce" if fb else "prepass", []).append(round(ms, 3))
            row["ran_first_bounce" if fb else "ran_prepass"] = (tb.GetOption("last_first_bounce"), tb.GetOption("last_primary_prepass"))
            if rep == 0: pics[fb] = tb.ReadAccumulation().copy()
    row["bit_identical"] = bool(np.array_equal(pics[0].view(np.uint32), pics[1].view(np.uint32)))
    row["rejects"] = tb.GetOption("debug_prepass_rejects") if False else None
    res[key] = row; print(key, row, flush=True)
    tb.SetOption("first_bounce", 0); tb.SetOption("overlap_launches", 1)
    if out_path: json.dump(res, open(out_path, "w"), indent=1)
