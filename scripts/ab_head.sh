#!/bin/bash
# same-box A/B of this tree against the committed tree unpacked and built under tracerboy_amd/_head/<name> (scripts/ab_rounds.py), preceded
# by the pre-pass parity tests of this tree;  bash scripts/ab_head.sh NAME [reps]  -> gpurun_out/r5/ab_<NAME>.json
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r5; mkdir -p $O
python3 -m pytest tests/test_primary_prepass.py tests/test_vw_van.py tests/test_gpu_parity.py -m gpu -k "not bench" -x -q 2>&1 | tail -3
python3 scripts/ab_rounds.py "$1" "${2:-3}" $O/ab_$1.json 2>&1 | tail -12
