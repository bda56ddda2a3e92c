#!/usr/bin/env python3
"""Per-kernel, per-REAL-launch averages of rocprofv3 counter_collection CSVs (shared by profile_bench.sh, pmc_quick.sh, pmc_mem.sh).

A frame-group / split-role launcher that is given zero frames (the host warming a stream up, context.cpp) still dispatches one
workgroup of the very kernel it warms, and that dispatch carries the kernel's name: averaged in, it understated every per-launch
counter by dispatches / real launches (C3 traffic 100 GB instead of 122 GB).  A dispatch of ONE workgroup
(Grid_Size == Workgroup_Size) is therefore not a launch; what remains is averaged, and the number of launches used is reported."""
import collections, csv, glob, re


def aggregate(pattern, keep, key=None):
    """pattern: glob of counter_collection.csv files; keep(kernel_name) -> bool; key(kernel_name) -> label (default: first 80 chars).
    Returns {label: {counter: mean over real launches, "dispatches": real launches, "warm_dispatches_dropped": n}}."""
    vals = collections.defaultdict(lambda: collections.defaultdict(dict))   # label -> counter -> dispatch -> value
    warm = collections.defaultdict(set)
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not keep(k): continue
            label = key(k) if key else k[:80]
            if int(r["Grid_Size"]) <= int(r["Workgroup_Size"]): warm[label].add((f, r["Dispatch_Id"])); continue
            d = vals[label][r["Counter_Name"]]
            d[(f, r["Dispatch_Id"])] = d.get((f, r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    out = {}
    for label, counters in vals.items():
        out[label] = {c: sum(d.values()) / max(len(d), 1) for c, d in counters.items()}
        out[label]["dispatches"] = max(len(d) for d in counters.values())
        out[label]["warm_dispatches_dropped"] = len(warm[label])
    return out


def pt_key(k):
    m = re.search(r"(pt_\w+|wf_\w+|accumulate_samples_kernel)<?([^>(]*)", k)
    return (m.group(1) + "<" + m.group(2) + ">") if m else k[:80]
