#!/bin/bash
# Launches in flight (option overlap_depth = 2 / 3 / 4): asynchronous steps of a rank of 8 and of the whole frame (scripts/async_rate.py).
#   gpurun_out/r6/overlap_depth.jsonl
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r6
OUT=gpurun_out/r6/overlap_depth.jsonl; rm -f $OUT
for d in 2 3 4; do
  for a in "vwvan --world 8 --spp 8,32" "c4 --world 8 --spp 8,32" "c5 --world 8 --spp 8,32" "c2 --world 8 --spp 64" "vwvan" "c4" "c5" "c3" "c2" "teapot" "vwvan_2level"; do
    timeout 300 python3 scripts/async_rate.py $a --steps 16 --opt overlap_depth=$d 2>/dev/null | grep "^{" | tee -a $OUT | cut -c1-220
  done
done
