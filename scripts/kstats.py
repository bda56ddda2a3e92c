#!/usr/bin/env python3
"""Prints calls and average duration of the path-tracing kernels in a rocprofv3 *kernel_stats.csv (names contain commas: csv module)."""
import csv, sys
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "pt_" in n or "accumulate" in n or "wf_" in n:
            print("%-60s calls %4s  avg %9.3f ms  total %9.3f ms" % (n.replace("(anonymous namespace)::", "").replace("void ", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
