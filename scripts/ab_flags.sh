#!/bin/bash
# Compiler-flag sweep on the GPU box: bash scripts/ab_flags.sh "<flags 1>" "<flags 2>" ...   (each built with TB_EXTRA_FLAGS, C2 + C3@32spp timed)
set -u
cd "$GRAFT_REPO_ROOT"
for FL in "$@"; do
  TB_EXTRA_FLAGS="$FL" timeout 600 python3 -m tracerboy_amd.build --force > /dev/null 2>&1 || { echo "build failed for [$FL]"; continue; }
  A=$(timeout 120 python3 bench.py --no-c3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  B=$(timeout 120 python3 bench.py --no-c3 --no-cpu-baseline --scene proc0:870000 --spp 32 --depth 6 --steps 3 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  C=$(timeout 120 python3 bench.py --no-c3 --no-cpu-baseline --scene proc2:2980000 --width 3840 --height 2160 --spp 4 --depth 16 --builder 4 --steps 3 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  D=$(timeout 120 python3 bench.py --no-c3 --no-cpu-baseline --scene proc1:700000 --width 3840 --height 2160 --spp 4 --depth 6 --builder 4 --steps 3 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  echo "[$FL] C2 $A  C3 $B  C5 $C  C4 $D"
done
TB_EXTRA_FLAGS="" python3 -m tracerboy_amd.build --force > /dev/null 2>&1
