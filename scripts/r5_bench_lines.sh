#!/bin/bash
# the three committed bench lines of a build: default, --steps 20 --warmup 5, and two ranks sharing one GPU (gloo)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r5; mkdir -p $O
( time python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python3 bench.py --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> /dev/null
TB_BENCH_SHARE_DEVICE=1 TB_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_n2_shared.json 2> $O/bench_n2_shared.err; echo "n2 rc $?"
python3 - <<'PY'
import json
for f in ('bench_default', 'bench_steps20_warmup5'):
    d = json.load(open('gpurun_out/r5/%s.json' % f))
    print(f, 'c2', d['value'], d['ms_per_step'], d['roofline'].get('frac'))
    for k in d:
        if k.startswith('roofline_'): print('  ', k, d[k]['value'], d[k]['ms_per_step'], d[k]['avg_launch_ms'], d[k]['frac'], d[k].get('vmem_spill_share'), d[k].get('pmc_stale'))
d = json.load(open('gpurun_out/r5/bench_n2_shared.json'))
print('n2', d['value'], [(k, d[k].get('value')) for k in d if k.startswith('scale_')])
PY
