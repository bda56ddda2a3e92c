#!/bin/bash
# After the costly-regions-first change: the GPU suite, the per-rank sweeps the scale legs' expected speed-ups are read from, the A/B of the option.
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r6; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; tail -3 $O/gpu_suite.log
python3 scripts/rank_imbalance.py $O/rank_imbalance.json > $O/rank_imbalance.log 2>&1; grep "world 8" $O/rank_imbalance.log
python3 scripts/rank_imbalance.py $O/rank_imbalance_32spp.json --spp 32 --steps 2 --worlds 1,8 > $O/rank_imbalance_32spp.log 2>&1; grep "world 8" $O/rank_imbalance_32spp.log
bash scripts/costly_first_ab.sh > /dev/null 2>&1; wc -l $O/costly_first.jsonl
