#!/bin/bash
# One lease's figures for the box-to-box spread: the default bench's workloads, same build, this box.
#   gpurun -- 'bash scripts/box_spread.sh N'   on several leases  ->  gpurun_out/r5/box_<N>.json; scripts/box_spread_table.py folds them
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r5; mkdir -p $O
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-readback > $O/box_$1.json 2> $O/box_$1.err
python3 - "$O/box_$1.json" <<'PY'
import json, sys, socket
d = json.load(open(sys.argv[1]))
print(socket.gethostname(), "c2", d["value"], {k[9:]: d[k]["value"] for k in d if k.startswith("roofline_")})
PY
