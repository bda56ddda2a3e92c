#!/usr/bin/env python3
"""A/B of the primary-visibility pre-pass (option primary_prepass, pt_scene.h TbDeviceTargets::primaryGeom): same scene, same frames,
with and without; the accumulation buffers must be equal bit for bit, the times are the median of `reps` synchronous renders.

  python scripts/prepass_ab.py [--out profiles/r3/prepass_ab.json]"""
import argparse, copy, json, os, sys, time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracerboy_amd import api  # noqa: E402


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--out", default=None); ap.add_argument("--reps", type=int, default=5); ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    tb = api.TracerBoy()
    s0 = api.GetDefaultOutputSettings(); s0.EnableBlueNoise = 0
    cases = [("c3 proc0:870000 1920x1080x32 depth6", (0, 870000, 1234), 1920, 1080, 32, 6),
             ("c4-class proc1:700000 3840x2160x8 depth6", (1, 700000, 1234), 3840, 2160, 8, 6),
             ("c5-class proc2:2980000 3840x2160x8 depth16", (2, 2980000, 1234), 3840, 2160, 8, 16)]
    if a.quick: cases = [("quick proc0:60000 640x360x16 depth6", (0, 60000, 3), 640, 360, 16, 6), ("quick proc1:60000 640x360x16 depth6", (1, 60000, 3), 640, 360, 16, 6)]
    rows = []
    for name, proc, W, H, F, depth in cases:
        s = copy.copy(s0); s.MaxBounces = depth
        tb.SetOption("bvh_builder", 4); tb.LoadProcedural(*proc); tb.SetOption("bvh_builder", 0)
        row = {"case": name}
        imgs = {}
        for layout in (0, 1):
            for pre in (0, 1):
                tb.SetOption("node_layout", layout); tb.SetOption("primary_prepass", 2 * pre)   # never / asked for (1 would be the try-and-keep default)
                ts = []
                for r in range(a.reps + 1):
                    tb.InvalidateHistory(); t = time.perf_counter(); tb.Render(W, H, F, s, 0.0); ts.append(time.perf_counter() - t)
                assert tb.GetOption("last_primary_prepass") == pre and tb.GetOption("last_node_layout") == layout, (tb.GetOption("last_primary_prepass"), tb.GetOption("last_node_layout"))
                imgs[(layout, pre)] = tb.ReadAccumulation()
                ms = float(np.median(ts[1:]) * 1e3)
                row["layout%s_prepass%d_ms" % ("BC"[layout], pre)] = round(ms, 3)
                row["layout%s_prepass%d_Msamples_s" % ("BC"[layout], pre)] = round(W * H * F / ms / 1e3, 1)
                row["variant"] = int(tb.GetOption("last_variant"))
        row["bit_identical_layoutB"] = bool(np.array_equal(imgs[(0, 0)].view(np.uint32), imgs[(0, 1)].view(np.uint32)))
        row["bit_identical_layoutC"] = bool(np.array_equal(imgs[(1, 0)].view(np.uint32), imgs[(1, 1)].view(np.uint32)))
        print(json.dumps(row), flush=True); rows.append(row)
    tb.SetOption("node_layout", 0); tb.SetOption("primary_prepass", 0)
    if a.out: json.dump(rows, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
