#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
for world in 1 8; do python3 scripts/rank_step.py c4 $world 0 2>/dev/null; done
python3 scripts/rank_step.py c4 8 0 --opt primary_prepass=2 2>/dev/null
python3 scripts/rank_step.py vwvan 8 0 2>/dev/null
python3 scripts/rank_step.py vwvan 8 2 2>/dev/null
python3 scripts/rank_step.py c5 8 0 2>/dev/null
python3 scripts/rank_step.py c2 8 0 2>/dev/null
python3 scripts/rank_step.py c2 1 0 2>/dev/null
python3 scripts/rank_step.py c3 1 0 --spp 16 2>/dev/null
python3 scripts/rank_step.py teapot 1 0 2>/dev/null
python3 -m pytest tests/test_buffer_reuse_stress.py tests/test_primary_prepass.py tests/test_vw_van.py -m gpu -x -q 2>&1 | tail -5
python3 bench.py --no-cpu-baseline > gpurun_out/r5/bench_3buf.json 2> gpurun_out/r5/bench_3buf.err; python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r5/bench_3buf.json'))
print('c2', d['value'], d['ms_per_step'])
for k in d:
    if k.startswith('roofline_'): print(k, d[k]['value'], d[k]['ms_per_step'], d[k]['avg_launch_ms'])
PY
bash scripts/rank_step_trace.sh c4_w8_3buf c4 8 0 | tail -32
