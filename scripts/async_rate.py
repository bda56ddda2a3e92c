#!/usr/bin/env python3
"""Steady-state rate of asynchronous steps (what bench.py times) for a workload at several samples-per-call and options:
does a launch cost more than its samples?   python scripts/async_rate.py LEG [--spp 64,128] [--steps 20] [--opt k=v ...] [--world N --rank r]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tracerboy_amd import api  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("leg"); ap.add_argument("--spp", default=None); ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--opt", action="append", default=[]); ap.add_argument("--world", type=int, default=1); ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--reps", type=int, default=3); ap.add_argument("--tile", type=int, default=0)   # 0: the leg's own tile size (bench.WORKLOADS)
a = ap.parse_args()
w = bench.WORKLOADS[a.leg]; W, H = w["W"], w["H"]
b = bench.Bench(api, 0); tb = b.tb; s = b.settings(w["depth"]); b.load_workload(a.leg)
tb.SetOption("overlap_launches", 2)
for kv in a.opt:
    k, v = kv.split("="); tb.SetOption(k, int(v))
tile = a.tile or w.get("tile", bench.TILE)
tb.SetTileAssignment(a.rank, a.world, tile, tile)
owned = tb.OwnedPixels(W, H)
for spp in [int(x) for x in (a.spp or str(w["spp"])).split(",")]:
    for _ in range(3):
        tb.InvalidateHistory(); tb.Render(W, H, spp, s, 0.0)
    best = None
    for rep in range(a.reps):
        tb.Sync(); t = time.perf_counter()
        for _ in range(a.steps):
            tb.InvalidateHistory(); tb.Render(W, H, spp, s, 0.0, sync=False)
        tb.Sync(); dt = (time.perf_counter() - t) / a.steps
        best = dt if best is None else min(best, dt)
    tb.InvalidateHistory(); tb.Render(W, H, spp, s, 0.0); sync_ms = tb.GetOption("last_kernel_us") / 1e3
    print(json.dumps({"leg": a.leg, "spp": spp, "opts": a.opt, "world": a.world, "rank": a.rank, "tile": tile, "async_ms_per_step": round(best * 1e3, 3), "Msamples_per_s": round(owned * spp / best / 1e6, 1),
                      "sync_kernel_ms": round(sync_ms, 3), "frames_per_launch": int(tb.GetOption("last_kernel_frames")), "frame_group": int(tb.GetOption("last_plan_frame_group"))}), flush=True)
