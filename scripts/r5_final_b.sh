#!/bin/bash
# round 5, final evidence of ONE build (digest printed first): spill share, stress record, per-rank sweeps at 8 and 32 spp, the bench lines
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r5; mkdir -p $O
python3 -c "import sys; sys.path.insert(0, '.'); from tracerboy_amd import build as b; print('kernel digest', b.kernel_digest())"
bash scripts/spill_share.sh 2>&1 | tail -6
python3 scripts/rank_imbalance.py $O/rank_imbalance.json > $O/rank_imbalance.log 2>&1; grep "world 8" $O/rank_imbalance.log
python3 scripts/rank_imbalance.py $O/rank_imbalance_32spp.json --spp 32 --steps 2 --worlds 1,8 > $O/rank_imbalance_32spp.log 2>&1; grep "world 8" $O/rank_imbalance_32spp.log
python3 scripts/rank_share_async.py 20 > $O/rank_share_async.json 2> /dev/null; cat $O/rank_share_async.json | cut -c1-300
( time python3 bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python3 bench.py --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> /dev/null
TB_BENCH_SHARE_DEVICE=1 TB_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_n2_shared.json 2> $O/bench_n2_shared.err; echo "n2 rc $?"
bash scripts/stress_r5.sh > $O/stress_summary.txt 2>&1; grep -c "bad 0" $O/stress_summary.txt; grep "wrong render\|digest" $O/stress_summary.txt
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r5/bench_default.json'))
print('c2', d['value'], d['ms_per_step'], d['roofline'].get('frac'))
for k in d:
    if k.startswith('roofline_'): print(k, d[k]['value'], d[k]['ms_per_step'], d[k]['avg_launch_ms'], d[k]['frac'], d[k].get('vmem_spill_share'))
PY
