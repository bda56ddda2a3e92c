#!/bin/bash
# storage order of the nodes (options node_order / node_order_top_levels, context_scene.cpp reorderNodes) on the SAH trees the legs build
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for o in "node_order=2,node_order_top_levels=10" "node_order=0" "node_order=1" "node_order=2,node_order_top_levels=6" "node_order=2,node_order_top_levels=14"; do
  echo "== $o"; TB_OPTS=$o python3 scripts/overlap_diag.py vwvan c4 c3 2>&1 | grep "sync ms" | cut -c1-60
done
