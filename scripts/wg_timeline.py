#!/usr/bin/env python3
"""Where does the end of a frame-group launch go?  Needs a library built with -DTB_WG_TIMELINE (scripts/build_variant.py tl --flags
-DTB_WG_TIMELINE --tus kernels): every workgroup of the lock-step kernel then leaves, in the tail of its slot-log row, when it started,
when its first lane found nothing left to draw and when each of its waves exited (100-MHz ticks).

    TB_LIB=tracerboy_amd/_sweep/libtracerboy_hip_tl.so python scripts/wg_timeline.py LEG [WORLD RANK] [--spp S] [--opt k=v ...] [--out f.json]

One launch at a time (synchronous renders: the timeline of a launch that has the chip to itself).  Prints / writes:
  launch_ms            first workgroup start -> last wave exit
  first_dry_ms         when the first lane anywhere found the lists empty, first_exit_ms when the first workgroup was gone
  tail_ms              launch_ms - first_dry_ms: how long the launch runs on with an emptying grid
  idle_lane_ms_share   sum over workgroups of (launch end - workgroup end) + (workgroup end - its first dry lane) / 2, over grid x launch:
                       an estimate of the lane-time the launch loses at its end (a dead workgroup's slot idles unless another launch takes it;
                       between a workgroup's first dead lane and its end about half its lanes idle)
  wg_dry_to_exit_ms    percentiles of (workgroup end - its first dry lane): the drain of ONE workgroup = its longest last paths
  wg_exit_spread_ms    percentiles of workgroup end times relative to the launch end: the ragged edge the item granularity leaves
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tracerboy_amd import api  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("leg")
ap.add_argument("world", type=int, nargs="?", default=1)
ap.add_argument("rank", type=int, nargs="?", default=0)
ap.add_argument("--spp", type=int, default=None)
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--out", default=None)
a = ap.parse_args()
w = bench.WORKLOADS[a.leg]
W, H, SPP = w["W"], w["H"], a.spp or w["spp"]
b = bench.Bench(api, 0)
tb = b.tb
s = b.settings(w["depth"])
b.load_workload(a.leg)
for kv in a.opt:
    k, v = kv.split("=")
    tb.SetOption(k, int(v))
tb.SetTileAssignment(a.rank, a.world, bench.TILE, bench.TILE)
hip = ctypes.CDLL("libamdhip64.so")
for _ in range(3):
    tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0)
rows = []
for rep in range(3):
    tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0)
    cap = tb.GetOption("debug_slot_log_cap")
    frames = tb.GetOption("last_kernel_frames")
    log = np.zeros((16 * 256, cap), np.uint64)
    rc = hip.hipMemcpy(ctypes.c_void_p(log.ctypes.data), ctypes.c_void_p(tb.GetOption("debug_slot_log_ptr")), ctypes.c_size_t(log.nbytes), 2)
    assert rc == 0, rc
    start, dry, drawn = log[:, cap - 1].astype(np.float64), log[:, cap - 2].astype(np.float64), log[:, cap - 7]
    wave_end = log[:, cap - 6:cap - 2].astype(np.float64)
    stat = log[:, cap - 11:cap - 7][:, ::-1].astype(np.float64)      # [-8] bind ticks, [-9] parked lane-trips, [-10] out-of-line resolves, [-11] binds
    live = (start > 0) & (wave_end.max(axis=1) > 0)
    start, dry, wave_end, drawn, stat = start[live], dry[live], wave_end[live], drawn[live], stat[live]
    end = wave_end.max(axis=1)
    t0, t1 = start.min(), end.max()
    tick = 1e-5  # ms per 100-MHz tick
    has_dry = dry > 0
    first_dry = (dry[has_dry].min() - t0) * tick if has_dry.any() else None
    d2e = (end[has_dry] - dry[has_dry]) * tick
    spread = (t1 - end) * tick
    launch = (t1 - t0) * tick
    # lane-time lost at the end: dead workgroups' slots until the launch ends (only workgroups that ended after the first dry lane count: the
    # ones before were replaced by the grid's second half) + half of each workgroup's own drain
    lost = np.where(end >= (t0 + (first_dry or 0) / tick), (t1 - end), 0.0).sum() * tick + 0.5 * d2e.sum()
    early = (start - t0) * tick < 0.05          # the resident half of the 2x grid; the rest start in slots the first leavers free and find the lists empty
    resident = int(early.sum())
    res_end = (t1 - end[early]) * tick
    res_drawn = drawn[early].astype(np.float64)
    body_rate = res_drawn.sum() / max(first_dry or launch, 1e-9)      # samples per ms while every lane had work (an upper bound: some were drawn later)
    pct = lambda v: [round(float(np.percentile(v, q)), 3) for q in (10, 50, 90, 99, 100)] if len(v) else None  # noqa: E731
    r = {"launch_ms": round(launch, 3), "kernel_ms_hip_events": round(tb.GetOption("last_kernel_us") / 1e3, 3), "frames_per_launch": int(frames),
         "workgroups": int(live.sum()), "resident_at_start": resident, "first_dry_ms": None if first_dry is None else round(first_dry, 3),
         "first_exit_ms": round(float((end.min() - t0) * tick), 3), "tail_ms": None if first_dry is None else round(launch - first_dry, 3),
         "idle_lane_ms_share": round(float(lost / max(resident, 1) / launch), 4),
         "wg_dry_to_exit_ms_p10_50_90_99_max": pct(d2e), "wg_exit_before_launch_end_ms_p10_50_90_99_max": pct(spread),
         "resident_exit_before_launch_end_ms_p10_50_90_99_max": pct(res_end), "resident_samples_min_p10_50_90_max": [float(res_drawn.min())] + pct(res_drawn)[:3] + [float(res_drawn.max())],
         "launch_if_no_tail_ms": round(float(res_drawn.sum() / body_rate), 3) if first_dry else None,
         "workgroups_started_late": int((~early).sum()),
         "binds_per_resident_wg": round(float(stat[early, 3].mean()), 1), "bind_us_mean": round(float(stat[early, 0].sum() / max(stat[early, 3].sum(), 1) * 0.01), 2),
         "bind_ms_per_resident_wg": round(float(stat[early, 0].mean() * tick), 3), "parked_lane_trips_per_resident_wg": round(float(stat[early, 1].mean()), 1),
         "slow_resolves_per_resident_wg": round(float(stat[early, 2].mean()), 1)}
    rows.append(r)
    print(json.dumps(r), flush=True)
doc = {"leg": a.leg, "world": a.world, "rank": a.rank, "spp": SPP, "opts": a.opt, "variant": bench.VARIANTS[tb.GetOption("last_variant")],
       "prepass": int(tb.GetOption("last_primary_prepass")), "frame_group": int(tb.GetOption("last_plan_frame_group")),
       "launches": rows}
if a.out:
    json.dump(doc, open(a.out, "w"), indent=1)
