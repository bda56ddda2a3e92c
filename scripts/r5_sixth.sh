#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r5; mkdir -p $O
python3 -m pytest tests/test_math.py tests/test_gpu_parity.py tests/test_material_branches.py -m gpu -x -q 2>&1 | tail -3
python3 scripts/ab_rounds.py r4 2 $O/ab_rounds.json > $O/ab_rounds.log 2>&1; tail -4 $O/ab_rounds.log | cut -c1-460
