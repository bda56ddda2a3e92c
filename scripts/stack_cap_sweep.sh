#!/bin/bash
# how many LDS stack entries do the 4K glass scenes need?  (stack_lds_cap = N: the rest of the stack in global memory)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for cap in 0 20 16 12 10 8; do
  echo "== stack_lds_cap $cap"; TB_OPTS="stack_lds_cap=$cap,stack_overflow_max=64" python3 scripts/overlap_diag.py c4 c5 2>&1 | grep "sync ms" | cut -c1-60
done
