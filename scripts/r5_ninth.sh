#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r5; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py tests/test_material_branches.py tests/test_vw_van.py tests/test_split_kernel.py tests/test_two_level.py -m gpu -x -q 2>&1 | tail -3
LEGS="c2 c3 c4 c5 teapot vwvan" PMC_LEGS="" bash scripts/ab_variants.sh rsel base rsel base rsel 2>&1 | grep "^base\|^rsel" | cut -c1-150
python3 scripts/ab_rounds.py r4 1 $O/ab_rounds_tmp.json 2>&1 | cut -c1-460
