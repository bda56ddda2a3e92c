#!/usr/bin/env python3
"""bench.py's roofline_<leg> figure (overlap settled by the library's own trial, asynchronous steps) with the frame-group size forced:
   python scripts/leg_group_sweep.py c3,teapot 0,8,16      (0 = the plan's own choice)"""
import copy, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch  # noqa: E402
import bench  # noqa: E402
from tracerboy_amd import api  # noqa: E402
bench.NO_PARITY = True
legs = sys.argv[1].split(","); groups = [int(x) for x in sys.argv[2].split(",")]
b = bench.Bench(api, 0)
base = copy.deepcopy(bench.WORKLOADS)
for leg in legs:
    for g in groups:
        bench.WORKLOADS[leg] = copy.deepcopy(base[leg]); bench.WORKLOADS[leg].setdefault("opts", {})["frame_group"] = g
        r = bench.extra_leg(b, np, torch, leg, 8)
        print(json.dumps({"leg": leg, "frame_group_option": g, "value": r["value"], "ms_per_step": r["ms_per_step"], "avg_launch_ms": r["avg_launch_ms"],
                          "launches_overlap": r["launches_overlap"], "prepass": r["primary_prepass"], "planned_group": int(b.tb.GetOption("last_plan_frame_group"))}), flush=True)
    b.tb.SetOption("frame_group", 0)
