#!/usr/bin/env python3
"""A/B of two builds of the library on one box, alternating processes: this tree against an earlier round's tree unpacked and built under
tracerboy_amd/_head/<name> (git archive <commit> tracerboy_amd include | tar -x -C tracerboy_amd/_head/<name>; its own build.py).
   python scripts/ab_rounds.py r3 [reps [out.json]]      -> Msamples/s per workload and build"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
CHILD = r'''
import json, os, sys, time
import numpy as np
from tracerboy_amd import api
root = sys.argv[1]
tb = api.TracerBoy(0)
out = {}
for key, scene, builder, W, H, F, D in (("c2", "cornell", 1, 1920, 1080, 64, 8), ("c3", "proc0:870000", 4, 1920, 1080, 32, 6), ("c4", "proc1:700000", 4, 3840, 2160, 8, 6), ("c5", "proc2:2980000", 4, 3840, 2160, 8, 16),
                                       ("teapot", "Teapot/scene.pbrt", 1, 1920, 1080, 16, 8), ("vwvan", "vw-van/vw-van.pbrt", 4, 3840, 2160, 8, 6)):
    tb.SetOption("bvh_builder", builder)
    if scene == "cornell": tb.LoadScene(os.path.join(root, "tests/golden/scenes/cornell-box/scene.pbrt"))
    elif scene.endswith(".pbrt"): tb.LoadScene(os.path.join(root, "tests/golden/scenes", scene))
    else:
        k, n = scene[4:].split(":"); tb.LoadProcedural(int(k), int(n), 1234)
    s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = D
    for _ in range(2): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0)
    ts = []
    for _ in range(5):
        tb.InvalidateHistory(); t = time.perf_counter(); tb.Render(W, H, F, s, 0.0); ts.append(time.perf_counter() - t)
    for _ in range(3):                      # let the library's overlap trial settle (bursts of asynchronous calls)
        for _ in range(5): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0, sync=False)
        tb.Sync()
    t = time.perf_counter()
    for _ in range(6): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0, sync=False)
    tb.Sync(); ta = (time.perf_counter() - t) / 6
    out[key] = {"sync": round(W * H * F / float(np.median(ts)) / 1e6, 1), "async": round(W * H * F / ta / 1e6, 1), "prepass": int(tb.GetOption("last_primary_prepass"))}
out["_library"] = os.path.relpath(api.LIB_PATH, root)
print(json.dumps(out))
'''
rows = []
for r in range(reps):
    for which in ("this", name):
        pp = ROOT if which == "this" else os.path.join(ROOT, "tracerboy_amd", "_head", name)
        # cwd = the tree whose package is meant: `python -c` puts the working directory FIRST on sys.path, ahead of PYTHONPATH -- run from the
        # repo root (as rounds 3 and 4 did) both children imported THIS tree's package and the "A/B" compared a build with itself, which is
        # why it "agreed within 0.5 %".  The child now reports which library it loaded.
        p = subprocess.run([sys.executable, "-c", CHILD, ROOT], capture_output=True, text=True, cwd=pp, env=dict(os.environ, PYTHONPATH=pp))
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if not line: print(which, "failed:", p.stderr[-500:]); continue
        print(which, line[-1], flush=True)
        rows.append({"build": which, "rep": r, **json.loads(line[-1])})
if len(sys.argv) > 3:
    json.dump({"against": name, "method": "alternating fresh processes on one box; sync = median of 5 synchronous renders, async = 6 renders enqueued back to back after the overlap trial settled; Msamples/s",
               "rows": rows}, open(sys.argv[3], "w"), indent=1)
