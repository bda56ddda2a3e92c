#!/bin/bash
# Msamples/s of the 4K glass scenes with the sss copy held to 4 / 5 / 6 waves per SIMD (libraries from scripts/build_sss_sweep.py), bursts of asynchronous renders
set -u
cd "$GRAFT_REPO_ROOT"
cp tracerboy_amd/libtracerboy_hip.so /tmp/lib_default.so
trap 'cp /tmp/lib_default.so tracerboy_amd/libtracerboy_hip.so' EXIT   # an interrupted run must not leave a sweep build in the tree
for W in "$@"; do
  cp tracerboy_amd/_sweep/libtracerboy_hip_sss$W.so tracerboy_amd/libtracerboy_hip.so
  python3 - "$W" <<'PY'
import sys, os, time, json
sys.path.insert(0, os.getcwd())
from tracerboy_amd import api
tb = api.TracerBoy(0)
out = {"waves": int(sys.argv[1])}
for key, scene, builder, W, H, F, D in (("c4 lbvh", "proc1:700000", 4, 3840, 2160, 8, 6), ("c4 sah", "proc1:700000", 1, 3840, 2160, 8, 6), ("c5 lbvh", "proc2:2980000", 4, 3840, 2160, 8, 16)):
    tb.SetOption("bvh_builder", builder); k, n = scene[4:].split(":"); tb.LoadProcedural(int(k), int(n), 1234)
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
    out[key] = {"msamples": round(best, 1), "split_stack_entries": tb.GetOption("last_plan_stack_overflow"), "overlap": tb.GetOption("last_overlap")}
print(json.dumps(out), flush=True)
PY
done
cp /tmp/lib_default.so tracerboy_amd/libtracerboy_hip.so
