#!/bin/bash
# round 5: the whole GPU suite, then the default bench line and the two-ranks-on-one-GPU line; logs under gpurun_out/r5
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r5; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -5 $O/gpu_suite.log
python3 __graft_entry__.py --smoke 2>&1 | tail -2
python3 bench.py --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench_steps20.err; echo "bench rc $?"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r5/bench_steps20_warmup5.json'))
print('c2', d['value'], d['ms_per_step'], d['roofline']['frac'])
for k in d:
    if k.startswith('roofline_'): print(k, d[k]['value'], d[k]['ms_per_step'], d[k]['avg_launch_ms'], d[k]['frac'], d[k]['kernel_variant'], d[k]['primary_prepass'], d[k]['launches_overlap'])
PY
