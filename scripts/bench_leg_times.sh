#!/bin/bash
# wall clock of the default bench.py run by leg
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
t() { local s=$(date +%s%N); "$@" > /dev/null 2>&1; local e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms : ${*:2}"; }
t python3 bench.py --no-legs
for l in c3 c4 c5 teapot vwvan vwvan_2level; do t python3 bench.py --no-cpu-baseline --no-readback --legs $l; done
t python3 bench.py
