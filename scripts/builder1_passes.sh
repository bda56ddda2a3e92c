#!/bin/bash
# builder 1 (SAH) with 0 / 1 reinsertion passes against builder 4 on the other workloads
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
for w in c3 c5 c4 vwvan_2level teapot; do
  for p in 0 1; do echo "== $w passes $p"; TB_REINSERT_PASSES=$p timeout 900 python3 scripts/vwvan_builders.py gpurun_out/r5/${w}_builder1_passes$p.json --builders 1 --workload $w 2>&1 | grep "^builder" | cut -c1-200; done
done
echo "== c3 builder 4"; timeout 600 python3 scripts/vwvan_builders.py --builders 4 --workload c3 2>&1 | grep "^builder" | cut -c1-200
echo "== c5 builder 4"; timeout 600 python3 scripts/vwvan_builders.py --builders 4 --workload c5 2>&1 | grep "^builder" | cut -c1-200
