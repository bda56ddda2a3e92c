#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
for lib in fbfree fb4 fb3; do echo "== $lib"; TB_LIB=$PWD/tracerboy_amd/_sweep/libtracerboy_hip_$lib.so python3 scripts/first_bounce_ab.py gpurun_out/r5/first_bounce_ab_$lib.json c4 vwvan c3 2>&1 | grep -v amdgpu.ids | cut -c1-250; done
# where does the time go: kernel trace of the first-bounce form on c4
cat > /tmp/fbtrace.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import bench
from tracerboy_amd import api
b = bench.Bench(api, 0); tb = b.tb; w = bench.WORKLOADS["c4"]; s = b.settings(w["depth"]); b.load_workload("c4")
for fb in (0, 1):
    tb.SetOption("first_bounce", fb)
    for _ in range(4): tb.InvalidateHistory(); tb.Render(w["W"], w["H"], w["spp"], s, 0.0)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5/fbtrace -o t -- python3 /tmp/fbtrace.py > /dev/null 2>&1
python3 scripts/kstats.py $(find gpurun_out/r5/fbtrace -name "*kernel_stats.csv") | cut -c1-160
