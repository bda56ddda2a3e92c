#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
python3 -c "import sys; sys.path.insert(0, '.'); from tracerboy_amd import build as b; print('kernel digest', b.kernel_digest())"
bash scripts/profile_all.sh 2>&1 | grep -v "^$" | cut -c1-300
