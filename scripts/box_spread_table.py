#!/usr/bin/env python3
"""Folds gpurun_out/r5/box_<n>.json (scripts/box_spread.sh, one per lease, same build) into profiles/rN/box_spread.json: per workload the
values, their mean and (max - min) / mean -- the spread below which a cross-round or cross-lease delta is not progress."""
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = {}
files = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "r5", "box_*.json")))
for f in files:
    try: d = json.load(open(f))
    except Exception: continue
    rows.setdefault("c2", []).append(d["value"])
    for k in d:
        if k.startswith("roofline_"): rows.setdefault(k[9:], []).append(d[k]["value"])
out = {"leases": len(files), "unit": "Msamples/s", "method": "python bench.py --steps 10 --warmup 3 on separate gpurun leases, one build",
       "workloads": {k: {"values": v, "mean": round(sum(v) / len(v), 1), "spread": round((max(v) - min(v)) / (sum(v) / len(v)), 4)} for k, v in rows.items()}}
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r5", "box_spread.json"), "w"), indent=1)
print(json.dumps(out["workloads"], indent=1))
