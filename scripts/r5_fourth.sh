#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
for ss in 2 3; do
python3 scripts/rank_step.py c4 8 0 --opt side_streams=$ss 2>/dev/null
python3 scripts/rank_step.py c4 1 0 --opt side_streams=$ss 2>/dev/null
python3 scripts/rank_step.py vwvan 8 0 --opt side_streams=$ss 2>/dev/null
python3 scripts/rank_step.py vwvan 1 0 --opt side_streams=$ss 2>/dev/null
python3 scripts/rank_step.py c5 8 0 --opt side_streams=$ss 2>/dev/null
python3 scripts/rank_step.py c5 1 0 --opt side_streams=$ss 2>/dev/null
python3 scripts/rank_step.py c2 8 0 --opt side_streams=$ss 2>/dev/null
python3 scripts/rank_step.py c2 1 0 --opt side_streams=$ss 2>/dev/null
python3 scripts/rank_step.py c3 1 0 --opt side_streams=$ss 2>/dev/null
python3 scripts/rank_step.py teapot 1 0 --opt side_streams=$ss 2>/dev/null
done
python3 -m pytest tests/test_buffer_reuse_stress.py tests/test_primary_prepass.py tests/test_vw_van.py tests/test_two_level.py tests/test_split_kernel.py -m gpu -x -q 2>&1 | tail -5
for lib in expreduce cheaprng; do echo $lib; TB_LIB=$PWD/tracerboy_amd/_sweep/libtracerboy_hip_$lib.so python3 scripts/rank_step.py c2 1 0 2>/dev/null; TB_LIB=$PWD/tracerboy_amd/_sweep/libtracerboy_hip_$lib.so python3 scripts/rank_step.py c3 1 0 2>/dev/null;  TB_LIB=$PWD/tracerboy_amd/_sweep/libtracerboy_hip_$lib.so python3 scripts/rank_step.py c4 1 0 2>/dev/null; done
