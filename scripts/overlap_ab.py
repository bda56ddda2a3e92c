#!/usr/bin/env python3
"""Do back-to-back asynchronous renders gain from running on the two side streams at once (option overlap_launches)?  C2 / C3 do (+9 %);
the 4K glass scenes lose.  Msamples/s of 6 async renders, overlap on / off, per workload.   python scripts/overlap_ab.py [out.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tracerboy_amd import api
tb = api.TracerBoy(0)
rows = []
SETS = {"first": (("c2", "cornell", 1, 1920, 1080, 64, 8), ("c3 x32", "proc0:870000", 4, 1920, 1080, 32, 6), ("c3 x128", "proc0:870000", 4, 1920, 1080, 128, 6),
                  ("c4 4K x8", "proc1:700000", 4, 3840, 2160, 8, 6), ("c4 4K x32", "proc1:700000", 4, 3840, 2160, 32, 6), ("c4 1080p x8", "proc1:700000", 4, 1920, 1080, 8, 6), ("c4 1080p x32", "proc1:700000", 4, 1920, 1080, 32, 6),
                  ("c4 640x360 x16", "proc1:700000", 4, 640, 360, 16, 6), ("c5 4K x8", "proc2:2980000", 4, 3840, 2160, 8, 16), ("c5 1080p x16", "proc2:2980000", 4, 1920, 1080, 16, 16),
                  ("teapot 1080p x16", "teapot", 1, 1920, 1080, 16, 8)),
        "second": (("teapot 640x360 x16", "teapot", 1, 640, 360, 16, 8), ("teapot 1080p x4", "teapot", 1, 1920, 1080, 4, 8), ("teapot 1080p x64", "teapot", 1, 1920, 1080, 64, 8), ("teapot 4K x4", "teapot", 1, 3840, 2160, 4, 8),
                   ("c4 1440p x8", "proc1:700000", 4, 2560, 1440, 8, 6), ("c4 4K x2", "proc1:700000", 4, 3840, 2160, 2, 6), ("c3 4K x8", "proc0:870000", 4, 3840, 2160, 8, 6), ("c2 4K x16", "cornell", 1, 3840, 2160, 16, 8),
                   ("vw-van 4K x8", "vwvan", 4, 3840, 2160, 8, 6), ("vw-van 1080p x8", "vwvan", 4, 1920, 1080, 8, 6))}
which = sys.argv[2] if len(sys.argv) > 2 else "first"
for key, scene, builder, W, H, F, D in SETS[which]:
    tb.SetOption("bvh_builder", builder)
    if scene == "cornell": tb.LoadScene(os.path.join(ROOT, "tests/golden/scenes/cornell-box/scene.pbrt"))
    elif scene == "teapot": tb.LoadScene(os.path.join(ROOT, "tests/golden/scenes/Teapot/scene.pbrt"))
    elif scene == "vwvan": tb.LoadScene(os.path.join(ROOT, "tests/golden/scenes/vw-van/vw-van.pbrt"))
    else:
        k, n = scene[4:].split(":"); tb.LoadProcedural(int(k), int(n), 1234)
    s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = D
    row = {"workload": key}
    for ov in (2, 0, 2, 0):
        tb.SetOption("overlap_launches", ov)
        for _ in range(2): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0)
        tb.Sync(); t = time.perf_counter()
        for _ in range(6): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0, sync=False)
        tb.Sync(); dt = (time.perf_counter() - t) / 6
        row.setdefault("overlap" if ov else "one at a time", []).append(round(W * H * F / dt / 1e6, 1))
    tb.SetOption("overlap_launches", 1)      # the default: tried where it is in doubt
    for _ in range(4):
        for _ in range(5): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0, sync=False)
        tb.Sync()
    t = time.perf_counter()
    for _ in range(6): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0, sync=False)
    tb.Sync(); dt = (time.perf_counter() - t) / 6
    row["default policy"] = round(W * H * F / dt / 1e6, 1); row["chose_overlap"] = bool(tb.GetOption("last_overlap")); row["trial_phase"] = tb.GetOption("overlap_trial_phase")
    row["variant"] = ["matte", "env", "surf", "vol", "full", "sss"][tb.GetOption("last_variant")]
    rows.append(row); print(json.dumps(row), flush=True)
tb.SetOption("overlap_launches", 1)
if len(sys.argv) > 1: json.dump(rows, open(sys.argv[1], "w"), indent=1)
