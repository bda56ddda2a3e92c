#!/usr/bin/env python3
"""Tile imbalance of the 8-GPU configurations, measured on ONE GPU: for each 4K scene bench.py's scale_<leg> renders
(the reference's vw-van, the van-class and the bistro-class stand-ins) this GPU plays rank r of N for EVERY r in turn and
runs the step bench.py runs at N > 1 -- asynchronous render of the rank's own tiles, device-side pack, a stream-ordered
consumer of the packed tiles (a device copy stands in for the xGMI gather) -- K steps enqueued back to back.

A step of the N-GPU job takes what its slowest rank takes: t(1) / max_r t(r of N) is the speed-up the tile split can give,
max / mean over the ranks is the imbalance SURVEY.md 8e names as the expected limiter (a car in the middle of a sky).
Where max / mean exceeds 1.05 at N = 8 the same is measured with 32x32 tiles.

    python scripts/rank_imbalance.py [out.json] [--steps K] [--legs vwvan,c4,c5] [--worlds 1,2,4,8] [--spp S]
bench.py's expected_speedup_leg() reads profiles/rN/rank_imbalance.json (the bench's own 8-spp step) and, beside it,
profiles/rN/rank_imbalance_32spp.json (--spp 32: a step a quarter of the way to the configurations' own 256 / 1024 spp).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from tracerboy_amd import api, tiles  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("out", nargs="?", default=None)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--legs", default="vwvan,c4,c5")
ap.add_argument("--worlds", default="1,2,4,8")
ap.add_argument("--spp", type=int, default=None)   # samples per pixel of a step (default: the workload's own, 8 for the 4K legs)
ap.add_argument("--imbalance-threshold", type=float, default=1.05)
ap.add_argument("--opt", action="append", default=[])   # library options for an A/B, e.g. --opt costly_first=0
ap.add_argument("--tiles", default="")   # e.g. "32,16": sweep these tile sizes too, whatever the imbalance (the largest world only)
args = ap.parse_args()
worlds = [int(x) for x in args.worlds.split(",")]

b = bench.Bench(api, 0)
tb = b.tb
lib_stream = torch.cuda.ExternalStream(tb.Stream())
result = {"_method": "one GPU as rank r of N for every r; async step (render own tiles + pack + stream-ordered consumer), "
                     "%d steps back to back, best of 2 bursts; overlap_launches = 2 as in bench.py's scale legs" % args.steps}


def rank_ms(W, H, SPP, s, rank, world, tile):
    tb.SetTileAssignment(rank, world, tile, tile)
    cap = max(tiles.packed_capacity(W, H, world, tile, tile), 1)
    packed = [torch.zeros((cap, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    sink = torch.zeros_like(packed[0])
    n = [0]

    def step():
        k = n[0] & 1
        n[0] += 1
        tb.InvalidateHistory()
        tb.Render(W, H, SPP, s, 0.0, sync=False)
        lib_stream.wait_stream(torch.cuda.current_stream())
        tb.PackOwnedTo(packed[k].data_ptr(), sync=False)
        torch.cuda.current_stream().wait_stream(lib_stream)
        sink.copy_(packed[k], non_blocking=True)

    for _ in range(3):
        step()
    tb.Sync()
    torch.cuda.synchronize()
    best = None
    for _ in range(2):
        t = time.perf_counter()
        for _ in range(args.steps):
            step()
        tb.Sync()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / args.steps * 1e3
        best = ms if best is None else min(best, ms)
    return best


def sweep(W, H, SPP, s, tile):
    rows = {}
    for world in worlds:
        per = [round(rank_ms(W, H, SPP, s, r, world, tile), 3) for r in range(world)]
        mean = sum(per) / len(per)
        rows["world%d" % world] = {"per_rank_ms": per, "max_ms": max(per), "mean_ms": round(mean, 3),
                                   "max_over_mean": round(max(per) / mean, 4), "slowest_rank": per.index(max(per))}
        print("   tile %d world %d: max %.3f mean %.3f max/mean %.4f" % (tile, world, max(per), mean, max(per) / mean), flush=True)
    one = rows.get("world1")
    if one:
        for world in worlds:
            r = rows["world%d" % world]
            r["speedup_vs_1gpu"] = round(one["max_ms"] / r["max_ms"], 3)
            r["efficiency"] = round(one["max_ms"] / r["max_ms"] / world, 4)
    return rows


for key in args.legs.split(","):
    w = bench.WORKLOADS[key]
    W, H, SPP = w["W"], w["H"], args.spp or w["spp"]
    s = b.settings(w["depth"])
    t0 = time.time()
    b.load_workload(key)
    tb.SetOption("overlap_launches", 2)
    for kv in args.opt: tb.SetOption(kv.split("=")[0], int(kv.split("=")[1]))
    print("%s: loaded in %.1f s" % (key, time.time() - t0), flush=True)
    tile = w.get("tile", bench.TILE)          # the deal bench.py's scale_<leg> uses for this workload
    rows = sweep(W, H, SPP, s, tile)
    entry = {"workload": "%s %dx%d %dspp depth%d" % (bench.scene_label(w["scene"]), W, H, SPP, w["depth"]), "tile": tile,
             "deal": "round-robin (tile t -> rank t % N)", "kernel_variant": bench.VARIANTS[tb.GetOption("last_variant")]}
    entry.update(rows)
    top = rows.get("world%d" % max(worlds))
    if args.tiles:
        keep = worlds
        worlds = [1, max(keep)] if 1 in keep else [max(keep)]
        for t in [int(x) for x in args.tiles.split(",")]:
            entry["tile%d" % t] = sweep(W, H, SPP, s, t)
        worlds = keep
    elif top and top["max_over_mean"] > args.imbalance_threshold and tile != 32:
        print("   max/mean %.3f > %.2f: trying 32x32 tiles" % (top["max_over_mean"], args.imbalance_threshold), flush=True)
        entry["tile32"] = sweep(W, H, SPP, s, 32)
    result[key] = entry
    tb.SetTileAssignment(0, 1)
    tb.SetOption("overlap_launches", 1)
    if args.out:
        json.dump(result, open(args.out, "w"), indent=1)
print(json.dumps(result))
