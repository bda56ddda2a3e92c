#!/bin/bash
# kernel trace of one workload under this tree's library and under tracerboy_amd/_head/<name>'s: bash scripts/trace_ab.sh c3 r5a
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r5/trace_ab; mkdir -p $O
K=$1; N=$2; M=${3:-sync}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/this_$K -o t -- python3 scripts/trace_workload.py $K 8 $M > $O/this_$K.log 2>&1
export TB_LIB=$GRAFT_REPO_ROOT/tracerboy_amd/_head/$N/tracerboy_amd/libtracerboy_hip.so
rocprofv3 --kernel-trace --stats --output-format csv -d $O/head_$K -o t -- python3 scripts/trace_workload.py $K 8 $M > $O/head_$K.log 2>&1
unset TB_LIB
for w in this head; do echo "== $w"; grep variant $O/${w}_$K.log; f=$(find $O/${w}_$K -name "*kernel_stats.csv" | head -1); head -6 "$f" | cut -c1-200; done
