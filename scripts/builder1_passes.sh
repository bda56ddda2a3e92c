#!/bin/bash
# builder 1 (SAH): reinsertion passes over the largest <share> percent of the subtrees, on every leg's scene
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
for w in ${WORKLOADS:-c4 vwvan c3 c5 vwvan_2level teapot}; do
  for combo in ${COMBOS:-"1:100 3:3 3:10 6:3"}; do p=${combo%%:*}; sh=${combo##*:}
    echo "== $w passes $p share $sh"; timeout 900 python3 scripts/vwvan_builders.py gpurun_out/r5/${w}_b1_p${p}_s${sh}.json --builders 1 --workload $w --passes $p --share $sh 2>&1 | grep "^builder" | cut -c1-200
  done
done
