#!/usr/bin/env python3
"""Frame-group size and pre-pass on the 4K glass scenes with the round-4 kernels (walk loops free of scratch, sss at 6 waves, vol at 4): Msamples/s of
bursts of six asynchronous renders for G = auto / 4 / 8 / 16 / 32 and pre-pass never / forced / default.   python scripts/fg_sweep_r4.py [out.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tracerboy_amd import api
tb = api.TracerBoy(0); rows = []
for key, scene, builder, W, H, F, D in (("c4 4K x8", "proc1:700000", 4, 3840, 2160, 8, 6), ("c4 4K x32", "proc1:700000", 4, 3840, 2160, 32, 6), ("c5 4K x8", "proc2:2980000", 4, 3840, 2160, 8, 16),
                                        ("vw-van 4K x8", "vwvan", 4, 3840, 2160, 8, 6)):
    tb.SetOption("bvh_builder", builder)
    if scene == "vwvan": tb.LoadScene(os.path.join(ROOT, "tests/golden/scenes/vw-van/vw-van.pbrt"))
    else: k, n = scene[4:].split(":"); tb.LoadProcedural(int(k), int(n), 1234)
    s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = D
    for G, pre in ((0, 1), (4, 1), (8, 1), (16, 1), (32, 1), (0, 0), (0, 2)):
        if G > F: continue
        tb.SetOption("frame_group", G); tb.SetOption("primary_prepass", pre)
        for _ in range(2): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0)
        for _ in range(3):
            for _ in range(5): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0, sync=False)
            tb.Sync()
        best = 0
        for _ in range(2):
            t = time.perf_counter()
            for _ in range(6): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0, sync=False)
            tb.Sync(); best = max(best, W * H * F * 6 / (time.perf_counter() - t) / 1e6)
        row = {"workload": key, "frame_group_option": G, "frame_group": tb.GetOption("last_plan_frame_group"), "prepass_option": pre, "prepass": tb.GetOption("last_primary_prepass"), "msamples": round(best, 1), "overlap": tb.GetOption("last_overlap")}
        rows.append(row); print(json.dumps(row), flush=True)
    tb.SetOption("frame_group", 0); tb.SetOption("primary_prepass", 1)
if len(sys.argv) > 1: json.dump(rows, open(sys.argv[1], "w"), indent=1)
