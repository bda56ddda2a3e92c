#!/bin/bash
# A/B of compile-time switches on the GPU box: bash scripts/ab_build.sh "<flags A>" "<flags B>" [bench args...]
# Rebuilds the library with TB_EXTRA_FLAGS=<flags>, runs bench.py twice (--no-c3 --no-cpu-baseline) and prints value / ms_per_step.
set -u
cd "$GRAFT_REPO_ROOT"
A=$1; B=$2; shift 2
for FL in "$A" "$B"; do
  TB_EXTRA_FLAGS="$FL" python3 -m tracerboy_amd.build --force > /dev/null 2>&1 || { echo "build failed for [$FL]"; continue; }
  for i in 1 2; do
    python3 bench.py --no-c3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$FL]', d['value'], 'Msamples/s', d['ms_per_step'], 'ms; launch', d['roofline'].get('avg_launch_ms'))"
  done
done
TB_EXTRA_FLAGS="" python3 -m tracerboy_amd.build --force > /dev/null 2>&1
