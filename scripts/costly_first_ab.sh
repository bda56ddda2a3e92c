#!/bin/bash
# Costly regions first (option costly_first 0 / 1; libraries built with -DTB_COSTLY_STEP=N as tracerboy_amd/_sweep/libtracerboy_hip_csN.so): launches that
# wait (sync_kernel_ms) and asynchronous steps of a rank of 8 and of the whole frame (scripts/async_rate.py).   gpurun_out/r6/costly_first.jsonl
#   bash scripts/costly_first_ab.sh [tag ...]      (tags of variant libraries to run, option on, beside the tree's own with the option off and on)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r6
OUT=gpurun_out/r6/costly_first.jsonl; rm -f $OUT
CFGS=("base 0" "base 1"); for t in "$@"; do CFGS+=("$t 1"); done
for rep in 1 2; do for cfg in "${CFGS[@]}"; do
  tag=${cfg% *}; o=${cfg#* }
  if [ $tag = base ]; then unset TB_LIB; else export TB_LIB=$PWD/tracerboy_amd/_sweep/libtracerboy_hip_$tag.so; fi
  for a in "vwvan --world 8 --spp 8,32" "c4 --world 8 --spp 8,32" "c5 --world 8 --spp 8,32" "vwvan" "c4" "c5" "vwvan_2level"; do
    timeout 300 python3 scripts/async_rate.py $a --steps 12 --opt costly_first=$o 2>/dev/null | grep "^{" | sed "s/^{/{\"lib\": \"$tag\", /" | tee -a $OUT | cut -c1-240
  done
done; done
