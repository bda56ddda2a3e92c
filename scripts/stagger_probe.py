#!/usr/bin/env python3
"""Two launches in flight on the 4K glass scenes end up running side by side (both start together, share the chip, end together: no tail is
hidden -- profiles/r4/c4_trace_overlap.csv).  Does the overlap pay once the launches are STAGGERED?  Bursts of N asynchronous renders,
overlap forced on, the second call of the burst held back by a host sleep of d ms (d = 0: the burst as the library runs it now), against
the same burst one launch at a time.   python scripts/stagger_probe.py [out.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tracerboy_amd import api
tb = api.TracerBoy(0)
rows = []
N = 10
for key, scene, builder, W, H, F, D in (("c4 4K x8", "proc1:700000", 4, 3840, 2160, 8, 6), ("c5 4K x8", "proc2:2980000", 4, 3840, 2160, 8, 16), ("teapot 1080p x16", "teapot", 1, 1920, 1080, 16, 8),
                                        ("teapot 1080p x64", "teapot", 1, 1920, 1080, 64, 8)):
    tb.SetOption("bvh_builder", builder)
    if scene == "teapot": tb.LoadScene(os.path.join(ROOT, "tests/golden/scenes/Teapot/scene.pbrt"))
    else:
        k, n = scene[4:].split(":"); tb.LoadProcedural(int(k), int(n), 1234)
    s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = D
    row = {"workload": key}

    def burst(delay_ms):
        tb.Sync(); t = time.perf_counter()
        for i in range(N):
            if i == 1 and delay_ms: time.sleep(delay_ms / 1e3)
            tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0, sync=False)
        tb.Sync(); return W * H * F * N / (time.perf_counter() - t) / 1e6

    tb.SetOption("overlap_launches", 0)
    for _ in range(2): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0)
    one = tb.GetOption("last_kernel_us") / 1e3
    row["launch_ms"] = round(one, 2)
    row["one at a time"] = [round(burst(0), 1) for _ in range(2)]
    tb.SetOption("overlap_launches", 2)
    for _ in range(2): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0)
    for frac in (0.0, 0.25, 0.5, 0.75, 0.9):
        row["overlap, 2nd call held %.0f%% of a launch" % (frac * 100)] = [round(burst(frac * one), 1) for _ in range(2)]
    rows.append(row); print(json.dumps(row), flush=True)
tb.SetOption("overlap_launches", 1)
if len(sys.argv) > 1: json.dump(rows, open(sys.argv[1], "w"), indent=1)
