#!/bin/bash
# Dynamic VALU instruction count of each phase of the C2 lock-step kernel: the shipped library and the pricing builds
# (scripts/build_variant.py dK --flags=-DTB_EXP_DOUBLE=K --tus kernels/pt_variant_matte5.hip, K = 1..8: one phase evaluated twice) run the
# same two synchronous C2 launches under rocprofv3 --pmc; SQ_INSTS_VALU(dK) - SQ_INSTS_VALU(base) = wave-level VALU instructions of phase K.
# Also: ms per launch without the profiler, and a hash of the picture (a pricing build must not change a bit).
#   bash scripts/c2_instruction_mix.sh [LEG]      -> gpurun_out/r6/mix_<LEG>/{pmc_<lib>/..., times.txt}, summary by scripts/c2_instruction_mix.py
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
LEG=${1:-c2}
OUT=gpurun_out/r6/mix_$LEG; rm -rf $OUT; mkdir -p $OUT
for lib in base d1 d2 d3 d4 d5 d6 d7 d8; do
  if [ "$lib" = base ]; then unset TB_LIB; else export TB_LIB=$PWD/tracerboy_amd/_sweep/libtracerboy_hip_$lib.so; [ -f "$TB_LIB" ] || continue; fi
  python3 scripts/mix_step.py $LEG --time 2>/dev/null | sed "s/^/$lib /" | tee -a $OUT/times.txt
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_$lib -o p -- python3 scripts/mix_step.py $LEG > /dev/null 2> $OUT/pmc_$lib.err
done
python3 scripts/c2_instruction_mix.py $OUT $LEG
