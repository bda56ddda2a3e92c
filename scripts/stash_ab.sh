#!/bin/bash
# LDS stash experiment (pt_persistent.inc TB_LDS_STASH): the shipped library and the variant, both with 12 stack entries in LDS
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for r in 1 2; do
for lib in "" "$GRAFT_REPO_ROOT/tracerboy_amd/_sweep/libtracerboy_hip_stash7.so" "$GRAFT_REPO_ROOT/tracerboy_amd/_sweep/libtracerboy_hip_stash13.so"; do
  echo "== ${lib:-base}"; if [ -n "$lib" ]; then export TB_LIB=$lib; else unset TB_LIB; fi
  TB_OPTS="stack_lds_cap=12,stack_overflow_max=64" python3 scripts/overlap_diag.py c4 c5 c3 teapot vwvan 2>&1 | grep "sync ms" | cut -c1-60
done; done
