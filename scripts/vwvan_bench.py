#!/usr/bin/env python3
"""The reference's vw-van scene (tests/golden/scenes/vw-van: configs[3] minus the absent body shell) at 3840x2160: Msamples/s flattened and
two-level (240 instances), pre-pass never / forced / default policy, and WHICH branch of the launch policy fired.
   python scripts/vwvan_bench.py [out.json]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tracerboy_amd import api
VW = os.path.join(ROOT, "tests", "golden", "scenes", "vw-van", "vw-van.pbrt")
RULES = {10: "no tuned copy", 11: "tuned copy fits", 12: "tuned copy, split stack", 13: "tree too deep", 14: "no room", 15: "full feature set for instances",
         20: "no pre-pass kernel", 21: "option off", 22: "forced", 23: "small call", 24: "environment-lit", 25: "glass among others", 26: "trial"}
tb = api.TracerBoy(0)
s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = 6
W, H, F = 3840, 2160, 8
if len(sys.argv) > 2: W, H, F = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
out = {"scene": "vw-van (reference Scenes/vw-van minus mesh_00125.ply, synthetic sky) %dx%dx%d depth %d" % (W, H, F, s.MaxBounces), "rows": []}
for flatten in (1, 0):
    tb.SetOption("flatten_instances", flatten); tb.SetOption("bvh_builder", 4)
    t = time.time(); tb.LoadScene(VW); load_s = time.time() - t
    info = tb.SceneInfo(); pictures = []
    for pre, name in ((0, "never"), (2, "forced"), (1, "default")):
        tb.SetOption("primary_prepass", pre); ts = []
        for r in range((8 if pre == 1 else 4) if flatten else 2):          # the default policy may try both ways over its first calls
            tb.InvalidateHistory(); t = time.perf_counter(); tb.Render(W, H, F, s, 0.0); ts.append(time.perf_counter() - t)
        pictures.append(tb.ReadAccumulation())
        row = {"instances": "flattened" if flatten else "two-level (240 instances)", "prepass_option": name, "Msamples_per_s": round(W * H * F / float(np.median(ts[-3:] if flatten else ts[-1:])) / 1e6, 2),
               "prepass_used": bool(tb.GetOption("last_primary_prepass")), "variant": ["matte", "env", "surf", "vol", "full", "sss"][tb.GetOption("last_variant")],
               "copy_rule": RULES.get(tb.GetOption("last_plan_rule_copy")), "prepass_rule": RULES.get(tb.GetOption("last_plan_rule_prepass")),
               "frame_group": tb.GetOption("last_plan_frame_group"), "stack_overflow_entries": tb.GetOption("last_plan_stack_overflow"),
               "triangles": int(info.numTriangles), "bvh_nodes": int(info.bvhNodesB), "bvh_depth": int(info.bvhMaxDepth), "load_s": round(load_s, 2)}
        out["rows"].append(row); print(json.dumps(row), flush=True)
        if len(sys.argv) > 1: json.dump(out, open(sys.argv[1], "w"), indent=1)
    out.setdefault("bit_identical_across_prepass", []).append(bool(all(np.array_equal(pictures[0].view(np.uint32), p.view(np.uint32)) for p in pictures[1:])))
tb.SetOption("primary_prepass", 1); tb.SetOption("flatten_instances", 1); tb.SetOption("bvh_builder", 0)
# the split-role kernel has no feature set with mix materials: pipeline 4 falls back to the lock-step kernel here
if len(sys.argv) > 1: json.dump(out, open(sys.argv[1], "w"), indent=1)
