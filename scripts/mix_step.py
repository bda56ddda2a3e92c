#!/usr/bin/env python3
"""Two synchronous launches of a bench workload (the program scripts/c2_instruction_mix.sh puts under rocprofv3 --pmc); --time: ms per launch
(HIP events, best of 5) and a hash of the accumulated picture."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tracerboy_amd import api  # noqa: E402
leg = sys.argv[1]; w = bench.WORKLOADS[leg]; W, H, SPP = w["W"], w["H"], w["spp"]
b = bench.Bench(api, 0); tb = b.tb; s = b.settings(w["depth"]); b.load_workload(leg)
tb.SetOption("overlap_launches", 0)
for kv in [a for a in sys.argv[2:] if "=" in a]:     # k=v: tb_set_option
    k, v = kv.split("="); tb.SetOption(k, int(v))
n = 7 if "--time" in sys.argv else 2
ms = []
for _ in range(n):
    tb.InvalidateHistory(); tb.Render(W, H, SPP, s, 0.0); ms.append(tb.GetOption("last_kernel_us") / 1e3)
if "--time" in sys.argv:
    print(json.dumps({"leg": leg, "kernel_ms": round(min(ms[2:]), 3), "picture_sha1": hashlib.sha1(tb.ReadAccumulation().tobytes()).hexdigest()[:16],
                      "variant": bench.VARIANTS[tb.GetOption("last_variant")], "frames_per_launch": int(tb.GetOption("last_kernel_frames"))}))
