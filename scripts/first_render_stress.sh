#!/bin/bash
# N fresh processes, each: load a glass scene, a few small renders with the pre-pass forced, every one against the oracle.
# (The first render of a process is the one that sizes buffers and scratch.)
N=${1:-40}; bad=0
for i in $(seq $N); do
  out=$(timeout 60 python scripts/lost_item_stress.py 4 ${2:-2} 2>&1 | grep "wrong pixels\|total bad")
  echo "$out" | grep -q "total bad 0" || { bad=$((bad+1)); echo "process $i: $out"; }
done
echo "processes with a wrong render: $bad of $N"
