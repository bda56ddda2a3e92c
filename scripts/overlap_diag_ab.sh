#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python3 scripts/overlap_diag.py c3 c4 c5 teapot vwvan c2 2>&1 | grep -v Warning
TB_LIB=$GRAFT_REPO_ROOT/tracerboy_amd/_head/$1/tracerboy_amd/libtracerboy_hip.so python3 scripts/overlap_diag.py c3 c4 c5 teapot vwvan c2 2>&1 | grep -v Warning
