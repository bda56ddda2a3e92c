#!/bin/bash
# Quick PMC comparison of one bench configuration: bash scripts/pmc_quick.sh <tag> <bench args...>
# Writes gpurun_out/pmc_<tag>.json with per-kernel averages of a few SQ counters (two passes, no trace domains).
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/pmcq_$TAG; rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-parity $*"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/a -o a -- python3 bench.py $ARGS > /dev/null 2> $OUT/a.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/b -o b -- python3 bench.py $ARGS > /dev/null 2> $OUT/b.err
python3 - "$OUT" "$TAG" <<'PY'
import sys, json
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
from pmc_aggregate import aggregate, pt_key
from tracerboy_amd import build as tb_build
out, tag = sys.argv[1], sys.argv[2]
res = aggregate(out + "/**/*counter_collection.csv", lambda k: ("pt_" in k or "wf_" in k) and "63u" not in k, pt_key)   # real launches only (pmc_aggregate.py)
for k, d in res.items():
    if d.get("SQ_ACTIVE_INST_VALU"): d["lane_util"] = d["SQ_THREAD_CYCLES_VALU"] / (d["SQ_ACTIVE_INST_VALU"] * 64)
res["_kernel_digest"] = {"digest": tb_build.kernel_digest()}
json.dump(res, open("gpurun_out/pmc_%s.json" % tag, "w"), indent=1)
for k, d in res.items():
    if not k.startswith("_"): print(k, {c: (round(v / 1e6, 1) if v > 1000 else round(v, 3)) for c, v in sorted(d.items())})
PY
