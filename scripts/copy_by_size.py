#!/usr/bin/env python3
"""Which copy of a feature set should a scene of a given size run -- the one held to an occupancy (sss: 6 waves per SIMD, registers in scratch) or the base
copy (3 waves, hardly any scratch)?  Teapot (126 k triangles, surf) prefers 3 waves; the 0.7 M / 3 M triangle glass scenes prefer 6.  Glass scenes of
20 k ... 700 k triangles, option high_occupancy 1 / 0, bursts of six asynchronous renders.   python scripts/copy_by_size.py [out.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tracerboy_amd import api
tb = api.TracerBoy(0); rows = []
for kind, n in ((1, 20000), (1, 60000), (1, 120000), (1, 200000), (1, 400000), (1, 700000), (2, 100000), (2, 400000), (0, 50000), (0, 200000)):
    for W, H, F, D in ((1920, 1080, 16, 6),):
        tb.SetOption("bvh_builder", 4); tb.LoadProcedural(kind, n, 1234)
        s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = D if kind != 2 else 16
        row = {"scene": "proc%d:%d" % (kind, n), "triangles": int(tb.SceneInfo().numTriangles), "frame": "%dx%dx%d" % (W, H, F)}
        for hi in (1, 0):
            tb.SetOption("high_occupancy", hi)
            for _ in range(2): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0)
            for _ in range(3):
                for _ in range(5): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0, sync=False)
                tb.Sync()
            best = 0
            for _ in range(2):
                t = time.perf_counter()
                for _ in range(6): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0, sync=False)
                tb.Sync(); best = max(best, W * H * F * 6 / (time.perf_counter() - t) / 1e6)
            row["occupancy copy" if hi else "base copy"] = round(best, 1)
            if hi: row["variant"] = ["matte", "env", "surf", "vol", "full", "sss"][tb.GetOption("last_variant")]; row["prepass"] = tb.GetOption("last_primary_prepass")
        tb.SetOption("high_occupancy", 1)
        rows.append(row); print(json.dumps(row), flush=True)
if len(sys.argv) > 1: json.dump(rows, open(sys.argv[1], "w"), indent=1)
