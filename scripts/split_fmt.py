#!/usr/bin/env python3
"""One line per result of scripts/split_sweep.py (stdin: its JSON lines)."""
import json, sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"): continue
    r = json.loads(line)
    if "error" in r: print("%-22s ERROR %s" % (r["label"], r["error"][:400])); continue
    s = "%-22s %8.1f Msamples/s %8.2f ms" % (r["label"], r["msamples_s"], r["ms"])
    if "async_msamples_s" in r: s += "  async %8.1f" % r["async_msamples_s"]
    if "bit_exact_vs_pipeline0" in r: s += "  exact=%d" % r["bit_exact_vs_pipeline0"]
    p = r.get("profile")
    if p:
        tw, sw = max(p["t_waves"], 1), max(p["s_waves"], 1)
        s += "  occ I %.2f L %.2f S %.2f  cyc/step %5.0f  Tsleep/wave %6.0f  S: rounds/wave %5.0f cyc/round %6.0f sleeps/round %4.1f  steps/ray %.1f" % (
            p["inner_occupancy"], p["leaf_occupancy"], p["shade_occupancy"], p["cycles_per_walk_step"], p["t_sleeps"] / tw, p["s_rounds"] / sw,
            p["s_cycles"] / max(p["s_rounds"], 1), p["s_sleeps"] / max(p["s_rounds"], 1), (p["t_inner_lanes"] + p["t_leaf_lanes"]) / max(p["s_rays"], 1))
    print(s)
