#!/usr/bin/env python3
"""The reference's vw-van (flattened) under the BVH builders the library has: load time, tree depth, box / triangle tests per sample
(a count_rays launch of the shipped kernels) and the rate of asynchronous 4K x 8 spp steps.  builder 1 = SAH + reinsertion (host), 3 = LBVH + treelet
passes (host), 4 = the same passes on the GPU (bench.py's choice for this leg), 2 = plain LBVH on the GPU.
    python scripts/vwvan_builders.py [out.json] [--builders 4,3,2,1] [--max-load-s 120]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tracerboy_amd import api  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("out", nargs="?"); ap.add_argument("--builders", default="4,3,2,1"); ap.add_argument("--workload", default="vwvan")
ap.add_argument("--passes", type=int, default=None); ap.add_argument("--share", type=int, default=None)   # options reinsertion_passes / reinsertion_share over the workload's own
a = ap.parse_args()
b = bench.Bench(api, 0); tb = b.tb
w = bench.WORKLOADS[a.workload]; W, H, F = w["W"], w["H"], w["spp"]; s = b.settings(w["depth"])
rows = {}
for builder in [int(x) for x in a.builders.split(",")]:
    opts = dict(w.get("opts") or {})
    if a.passes is not None: opts["reinsertion_passes"] = a.passes
    if a.share is not None: opts["reinsertion_share"] = a.share
    t0 = time.time(); b.load(w["scene"], builder, opts); load_s = time.time() - t0
    for _ in range(2): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0)
    for _ in range(3):
        for _ in range(5): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0, sync=False)
        tb.Sync()
    t = time.perf_counter()
    for _ in range(6): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0, sync=False)
    tb.Sync(); ms = (time.perf_counter() - t) / 6 * 1e3
    variant, prepass = bench.VARIANTS[tb.GetOption("last_variant")], int(tb.GetOption("last_primary_prepass"))
    tb.SetOption("count_rays", 1); tb.InvalidateHistory(); tb.Render(W, H, 1, s, 0.0); st = tb.ReadbackStats().rays; tb.SetOption("count_rays", 0); tb.InvalidateHistory()
    n = float(st.samples)
    rows["builder%d" % builder] = {"load_s": round(load_s, 2), "ms_per_step": round(ms, 2), "Msamples_per_s": round(W * H * F / ms / 1e3, 1),
                                   "boxes_per_sample": round(st.boxesTested / n, 2), "tris_per_sample": round(st.trianglesTested / n, 2),
                                   "rays_per_sample": round(st.rays / n, 3), "stack_overflow_entries": int(tb.GetOption("last_plan_stack_overflow")),
                                   "kernel_variant": variant, "prepass": prepass,
                                   "reinsertion_passes": opts.get("reinsertion_passes", "library default") if builder == 1 else None,
                                   "reinsertion_share": opts.get("reinsertion_share", 100) if builder == 1 else None}
    print("builder", builder, rows["builder%d" % builder], flush=True)
    if a.out: json.dump({"workload": a.workload, "rows": rows}, open(a.out, "w"), indent=1)
