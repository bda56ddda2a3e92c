#!/bin/bash
# the headline's 36-triangle tree under builder 1 with 0 ... 32 reinsertion passes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
for p in 16 0 1 2 4 8 32; do echo "== passes $p"; timeout 300 python3 scripts/vwvan_builders.py --builders 1 --workload c2 --passes $p 2>&1 | grep "^builder" | cut -c1-170; done
