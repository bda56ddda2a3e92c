#!/bin/bash
# the committed bench lines, then the whole GPU suite + smoke
cd "$GRAFT_REPO_ROOT"; bash scripts/r5_bench_lines.sh
export TMPDIR=/tmp; O=gpurun_out/r5
python3 -m pytest tests -m gpu -q > $O/gpu_suite.log 2>&1; echo "suite rc $?"; tail -3 $O/gpu_suite.log
python3 __graft_entry__.py --smoke 2>&1 | tail -1
