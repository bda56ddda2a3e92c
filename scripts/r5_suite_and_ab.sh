#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r5; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -3 $O/gpu_suite.log
python3 scripts/ab_rounds.py r4 3 $O/ab_rounds.json > $O/ab_rounds.log 2>&1; tail -6 $O/ab_rounds.log | cut -c1-460
