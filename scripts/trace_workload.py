#!/usr/bin/env python3
"""A few synchronous renders of one bench workload, for a kernel trace:  rocprofv3 --kernel-trace --stats -d DIR -- python3 scripts/trace_workload.py c3 [reps]
(TB_LIB selects the library, tracerboy_amd/api.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tracerboy_amd import api  # noqa: E402
key = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
asynchronous = len(sys.argv) > 3 and sys.argv[3] == "async"   # the renders enqueued back to back, one wait at the end
b = bench.Bench(api, 0)
w = bench.WORKLOADS[key]
b.load_workload(key)
s = b.settings(w["depth"])
for _ in range(reps):
    b.tb.InvalidateHistory(); b.tb.Render(w["W"], w["H"], w["spp"], s, 0.0, sync=not asynchronous)
b.tb.Sync()
print(key, "variant", b.tb.GetOption("last_variant"), "prepass", b.tb.GetOption("last_primary_prepass"), "compact", b.tb.GetOption("last_compact_hits"))
