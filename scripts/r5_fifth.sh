#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
for cap in 0 20 16 12 8; do
python3 scripts/rank_step.py c4 8 0 --opt side_streams=2 --opt stack_lds_cap=$cap 2>/dev/null
python3 scripts/rank_step.py c4 1 0 --opt side_streams=2 --opt stack_lds_cap=$cap 2>/dev/null
done
python3 scripts/rank_step.py vwvan 8 0 --opt side_streams=2 --opt stack_lds_cap=12 2>/dev/null
python3 scripts/rank_step.py c5 8 0 --opt side_streams=2 --opt stack_lds_cap=12 2>/dev/null
for lib in doublerng; do echo $lib; for leg in c2 c3 c4; do TB_LIB=$PWD/tracerboy_amd/_sweep/libtracerboy_hip_$lib.so python3 scripts/rank_step.py $leg 1 0 --opt side_streams=2 2>/dev/null; done; done
