#!/usr/bin/env python3
"""Copies what scripts/profile_all.sh left under gpurun_out/ into profiles/rN/ under the names bench.py reads:
   <tag>_kernel_stats.csv, <tag>_pmc_summary.json, <tag>_mem_counters.json, <tag>_bench.json.    python scripts/collect_profiles.py r4 [tags...]"""
import os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]; tags = sys.argv[2:] or ["c2", "c3", "c4", "c5", "teapot", "vwvan", "vwvan_2level"]
dst = os.path.join(ROOT, "profiles", rnd); os.makedirs(dst, exist_ok=True)
for t in tags:
    for src, name in ((os.path.join("gpurun_out", t, "kernel_stats.csv"), t + "_kernel_stats.csv"), (os.path.join("gpurun_out", t, "pmc_summary.json"), t + "_pmc_summary.json"),
                      (os.path.join("gpurun_out", t, "bench.json"), t + "_bench.json"), (os.path.join("gpurun_out", "pmcmem_%s.json" % t), t + "_mem_counters.json")):
        p = os.path.join(ROOT, src)
        if os.path.exists(p) and os.path.getsize(p) > 0: shutil.copy2(p, os.path.join(dst, name)); print("copied", name)
        else: print("MISSING", src)
