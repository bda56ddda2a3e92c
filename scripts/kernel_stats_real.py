#!/usr/bin/env python3
"""Kernel time per REAL launch from a rocprofv3 --kernel-trace CSV.

rocprofv3's own *kernel_stats.csv averages over every dispatch that carries a kernel's name -- also the zero-frame dispatches the host
warms its side streams with (one workgroup that finds its list empty, 13 us), which depressed `AverageNs` of the lock-step kernel by
the share of such dispatches (C3, round 5: 36.5 ms quoted where Total / real launches = 45.6 ms).  The rule is pmc_aggregate.py's: a
dispatch of ONE workgroup (Grid_Size <= Workgroup_Size) is not a launch.

    python scripts/kernel_stats_real.py <dir with *kernel_trace.csv> out.csv
writes Name, Calls (real), TotalDurationNs, AverageNs (= Total / real launches), MinNs, MaxNs, WarmDispatchesDropped."""
import collections, csv, glob, os, sys


def real_stats(trace_dir):
    rows = collections.defaultdict(list); warm = collections.Counter()
    for f in glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name") or r.get("Name")
            grid = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0); wg = int(r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or 1)
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            if grid <= wg and ("pt_" in name or "wf_" in name): warm[name] += 1; continue
            rows[name].append(dur)
    out = []
    for name, d in rows.items():
        out.append({"Name": name, "Calls": len(d), "TotalDurationNs": sum(d), "AverageNs": round(sum(d) / len(d), 1), "MinNs": min(d), "MaxNs": max(d),
                    "WarmDispatchesDropped": warm.get(name, 0)})
    out.sort(key=lambda r: -r["TotalDurationNs"])
    return out


if __name__ == "__main__":
    stats = real_stats(sys.argv[1])
    with open(sys.argv[2], "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "WarmDispatchesDropped"])
        w.writeheader(); w.writerows(stats)
    for r in stats[:6]:
        print("%-70s real launches %3d  avg %9.3f ms  (dropped %d one-workgroup dispatches)" % (r["Name"].replace("(anonymous namespace)::", "")[:70], r["Calls"], r["AverageNs"] / 1e6, r["WarmDispatchesDropped"]))
