#!/usr/bin/env python3
"""Does a workload's asynchronous stream of renders overlap its launches?  python scripts/overlap_diag.py c3 [c4 ...]   (TB_LIB selects the library)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tracerboy_amd import api  # noqa: E402
b = bench.Bench(api, 0); tb = b.tb
for key in sys.argv[1:]:
    w = bench.WORKLOADS[key]
    for kv in os.environ.get("TB_OPTS", "").split(","):   # e.g. TB_OPTS=node_order=1,overlap_launches=2 (set before the load: some act there)
        if kv: tb.SetOption(kv.split("=")[0], int(kv.split("=")[1]))
    b.load_workload(key)
    s = b.settings(w["depth"]); W, H, F = w["W"], w["H"], w["spp"]
    for _ in range(2): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0)
    t = time.perf_counter()
    for _ in range(4): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0)
    sync = (time.perf_counter() - t) / 4
    for _ in range(3):
        for _ in range(5): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0, sync=False)
        tb.Sync()
    t = time.perf_counter()
    for _ in range(6): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0, sync=False)
    tb.Sync(); asy = (time.perf_counter() - t) / 6
    print(key, "sync ms %.2f async ms %.2f" % (sync * 1e3, asy * 1e3), {k: tb.GetOption(k) for k in ("last_overlap", "overlap_trial_phase", "overlap_trial_us_overlapped",
          "overlap_trial_us_one_at_a_time", "last_primary_prepass", "last_compact_hits", "last_plan_rule_prepass", "last_plan_frame_group", "last_plan_stack_overflow")}, flush=True)
