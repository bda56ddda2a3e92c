#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r5; mkdir -p $O
python3 scripts/ab_rounds.py r4 2 $O/ab_rounds.json > $O/ab_rounds.log 2>&1; tail -4 $O/ab_rounds.log | cut -c1-400
bash scripts/spill_share.sh 2>&1 | tail -6
bash scripts/stress_r5.sh > $O/stress_summary.txt 2>&1; cat $O/stress_summary.txt | cut -c1-200
