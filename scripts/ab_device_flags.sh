#!/bin/bash
# A/B on the GPU box: the library built with / without extra device flags, the 4K glass scenes with both builders (deep tree -> split stack, shallow -> not)
set -u
cd "$GRAFT_REPO_ROOT"
for FL in "$@"; do
  TB_EXTRA_FLAGS="$FL" timeout 900 python3 -m tracerboy_amd.build --force > /dev/null 2>&1 || { echo "build failed for [$FL]"; continue; }
  python3 - "$FL" <<'PY'
import sys, os, time, json
sys.path.insert(0, os.getcwd())
from tracerboy_amd import api
tb = api.TracerBoy(0)
out = {"flags": sys.argv[1]}
SETS = {"sss": (("c4 lbvh", "proc1:700000", 4, 3840, 2160, 8, 6), ("c4 sah", "proc1:700000", 1, 3840, 2160, 8, 6), ("c5 lbvh", "proc2:2980000", 4, 3840, 2160, 8, 16), ("c4 200k sah", "proc1:200000", 1, 3840, 2160, 8, 6)),
        "rest": (("c2", "cornell-box/scene.pbrt", 1, 1920, 1080, 64, 8), ("c3", "proc0:870000", 4, 1920, 1080, 128, 6), ("teapot", "Teapot/scene.pbrt", 1, 1920, 1080, 16, 8), ("vw-van", "vw-van/vw-van.pbrt", 4, 3840, 2160, 8, 6),
                 ("c3 4K", "proc0:870000", 4, 3840, 2160, 8, 6)),
        "teapot": (("teapot x16", "Teapot/scene.pbrt", 1, 1920, 1080, 16, 8), ("teapot x64", "Teapot/scene.pbrt", 1, 1920, 1080, 64, 8), ("teapot 4K x4", "Teapot/scene.pbrt", 1, 3840, 2160, 4, 8), ("teapot 640x360 x16", "Teapot/scene.pbrt", 1, 640, 360, 16, 8)),
        "van": (("vw-van", "vw-van/vw-van.pbrt", 4, 3840, 2160, 8, 6), ("vw-van two-level", "vw-van/vw-van.pbrt", 4, 3840, 2160, 8, 6), ("vw-van 1080p", "vw-van/vw-van.pbrt", 4, 1920, 1080, 8, 6),
                ("c4 lbvh", "proc1:700000", 4, 3840, 2160, 8, 6), ("c5 lbvh", "proc2:2980000", 4, 3840, 2160, 8, 16))}
for key, scene, builder, W, H, F, D in SETS[os.environ.get("AB_SET", "sss")]:
    tb.SetOption("bvh_builder", builder); tb.SetOption("flatten_instances", 0 if "two-level" in key else 1)
    if scene.startswith("proc"): k, n = scene[4:].split(":"); tb.LoadProcedural(int(k), int(n), 1234)
    else: tb.LoadScene(os.path.join("tests", "golden", "scenes", scene))
    s = api.GetDefaultOutputSettings(); s.EnableBlueNoise = 0; s.MaxBounces = D
    for _ in range(2): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0)
    for _ in range(3):
        for _ in range(5): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0, sync=False)
        tb.Sync()
    best = 0
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(6): tb.InvalidateHistory(); tb.Render(W, H, F, s, 0.0, sync=False)
        tb.Sync(); best = max(best, W * H * F * 6 / (time.perf_counter() - t) / 1e6)
    out[key] = {"msamples": round(best, 1), "variant": tb.GetOption("last_variant"), "split_stack_entries": tb.GetOption("last_plan_stack_overflow"), "prepass": tb.GetOption("last_primary_prepass"), "overlap": tb.GetOption("last_overlap")}
print(json.dumps(out), flush=True)
PY
done
TB_EXTRA_FLAGS="" python3 -m tracerboy_amd.build --force > /dev/null 2>&1
