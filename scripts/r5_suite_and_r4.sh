#!/bin/bash
# the whole GPU suite + smoke of this tree, then its same-box A/B against the round-4 tree
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r5; mkdir -p $O
python3 -m pytest tests -m gpu -q > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -3 $O/gpu_suite.log
python3 __graft_entry__.py --smoke 2>&1 | tail -1
python3 scripts/ab_rounds.py r4 3 $O/ab_rounds_final.json > /dev/null 2>&1; echo "ab rc $?"
