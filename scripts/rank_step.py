#!/usr/bin/env python3
"""One GPU as rank r of N on one of bench.py's workloads: K asynchronous steps (render own tiles + pack + stream-ordered
consumer) back to back, printed as ms per step -- the program scripts/rank_step_trace.sh puts under rocprofv3 --kernel-trace to see
where a rank's step goes (pre-pass / lock-step kernel / fold / pack, gaps between them, what overlaps).
   python scripts/rank_step.py LEG WORLD RANK [--steps K] [--spp S] [--tile T] [--opt key=int ...]"""
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
ap.add_argument("leg")
ap.add_argument("world", type=int)
ap.add_argument("rank", type=int)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--spp", type=int, default=None)
ap.add_argument("--tile", type=int, default=bench.TILE)
ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
w = bench.WORKLOADS[a.leg]
W, H, SPP = w["W"], w["H"], a.spp or w["spp"]
b = bench.Bench(api, 0)
tb = b.tb
s = b.settings(w["depth"])
b.load_workload(a.leg)
tb.SetOption("overlap_launches", 2)
for kv in a.opt:
    k, v = kv.split("=")
    tb.SetOption(k, int(v))
tb.SetTileAssignment(a.rank, a.world, a.tile, a.tile)
cap = max(tiles.packed_capacity(W, H, a.world, a.tile, a.tile), 1)
packed = [torch.zeros((cap, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
sink = torch.zeros_like(packed[0])
lib_stream = torch.cuda.ExternalStream(tb.Stream())
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
out = []
for _ in range(2):
    t = time.perf_counter()
    for _ in range(a.steps):
        step()
    tb.Sync()
    torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t) / a.steps * 1e3, 3))
print(json.dumps({"leg": a.leg, "world": a.world, "rank": a.rank, "spp": SPP, "tile": a.tile, "ms_per_step": out,
                  "Msamples_per_s_rank": round(tb.OwnedPixels(W, H) * SPP / min(out) / 1e3, 1),
                  "frame_group": tb.GetOption("last_plan_frame_group"), "prepass": tb.GetOption("last_primary_prepass"),
                  "overlap": tb.GetOption("last_overlap"), "opts": a.opt}))
