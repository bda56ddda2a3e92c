#!/bin/bash
# same-box A/B of this tree against the round-4 tree under tracerboy_amd/_head/r4 (scripts/ab_rounds.py) -> gpurun_out/r5/ab_rounds_final.json
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
python3 scripts/ab_rounds.py r4 3 gpurun_out/r5/ab_rounds_final.json 2>&1 | tail -3 | cut -c1-200
