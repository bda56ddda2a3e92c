#!/bin/bash
# round 5, first GPU call: the restructured bench.py at N = 1 and as two ranks on one GPU, then the per-rank imbalance sweep
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r5; mkdir -p $O
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; tail -c 600 $O/bench_default.err
TB_BENCH_SHARE_DEVICE=1 TB_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_n2_shared.json 2> $O/bench_n2_shared.err; echo "n2 rc $?"; tail -c 600 $O/bench_n2_shared.err
python3 scripts/rank_imbalance.py $O/rank_imbalance.json > $O/rank_imbalance.log 2>&1; echo "imb rc $?"; tail -30 $O/rank_imbalance.log | cut -c1-300
