#!/usr/bin/env python3
"""Pipeline 4 (split-role kernel, pt_split.inc) against pipeline 0 on one workload, over workgroup shapes and thresholds.

   python scripts/split_sweep.py --scene cornell-box --spp 64 --depth 8 --configs 4x4,4x8,6x10 [--ready 32 --refill 16] [--out file.json]
Each config: TxS[:ready[:refill[:fg[:cap]]]].  Prints Msamples/s (median of --reps synchronous renders after a warm-up) and checks the picture
against pipeline 0's bit for bit."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tracerboy_amd import api


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cornell-box"); ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64); ap.add_argument("--depth", type=int, default=8); ap.add_argument("--builder", type=int, default=1)
    ap.add_argument("--configs", default="4x4"); ap.add_argument("--reps", type=int, default=3); ap.add_argument("--out", default="")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT", help="extra options for every split config")
    ap.add_argument("--profile", action="store_true", help="one more render per config with the counting copy: occupancies, sleeps, cycles per step")
    ap.add_argument("--async-steps", type=int, default=0, help="also time N back-to-back async renders (overlapped launches)")
    a = ap.parse_args()
    tb = api.TracerBoy(0)
    tb.SetOption("bvh_builder", a.builder)
    if a.scene == "cornell-box": tb.LoadScene(os.path.join(ROOT, "tests", "golden", "scenes", "cornell-box", "scene.pbrt"))
    elif a.scene.startswith("proc"):
        k, n = a.scene[4:].split(":"); tb.LoadProcedural(int(k), int(n), 1234)
    else: tb.LoadScene(a.scene)
    s = api.GetDefaultOutputSettings(); s.MaxBounces = a.depth
    W, H, SPP = a.width, a.height, a.spp

    def timed(label):
        tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0)   # warm-up
        ts = []
        for _ in range(a.reps):
            tb.InvalidateHistory(); t = time.perf_counter(); tb.Render(W, H, SPP, s, 0.0); ts.append(time.perf_counter() - t)
        r = {"label": label, "ms": round(float(np.median(ts)) * 1e3, 3), "msamples_s": round(W * H * SPP / float(np.median(ts)) / 1e6, 1), "pipeline": tb.GetOption("last_pipeline")}
        if a.async_steps:
            tb.InvalidateHistory(); t = time.perf_counter()
            for _ in range(a.async_steps): tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0, sync=False)
            tb.Sync(); dt = time.perf_counter() - t
            r["async_msamples_s"] = round(W * H * SPP * a.async_steps / dt / 1e6, 1)
        return r

    res = []
    tb.SetOption("pipeline", 0)
    r = timed("pipeline0"); ref = tb.ReadAccumulation().copy(); res.append(r); print(json.dumps(r), flush=True)
    tb.SetOption("pipeline", 4)
    for kv in a.opt:
        k, v = kv.split("="); tb.SetOption(k, int(v))
    for cfg in a.configs.split(","):
        parts = cfg.split(":"); t, sh = parts[0].split("x")
        vals = [int(x) for x in parts[1:]] + [None] * 4
        tb.SetOption("split_trav", int(t)); tb.SetOption("split_shade", int(sh))
        tb.SetOption("split_ready", vals[0] if vals[0] is not None else 32); tb.SetOption("split_refill", vals[1] if vals[1] is not None else 16)
        tb.SetOption("split_frame_group", vals[2] if vals[2] is not None else 8); tb.SetOption("split_stack_cap", vals[3] if vals[3] is not None else 0)
        try:
            r = timed(cfg)
            r["bit_exact_vs_pipeline0"] = bool(np.array_equal(tb.ReadAccumulation().view(np.uint32), ref.view(np.uint32)))
            if a.profile:
                tb.SetOption("split_profile", 1); tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0)
                p = tb.SplitProfile(); tb.SetOption("split_profile", 0)
                r["profile"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in p.items()}
        except Exception as e:
            r = {"label": cfg, "error": str(e)[:500]}
        res.append(r); print(json.dumps(r), flush=True)
    if a.out:
        json.dump({"scene": a.scene, "W": W, "H": H, "spp": SPP, "depth": a.depth, "results": res}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
