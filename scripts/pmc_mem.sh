#!/bin/bash
# Memory-pipeline counters of one bench configuration (texture addresser, L1, address translation, VMEM issue):
#   bash scripts/pmc_mem.sh <tag> <bench args...>   ->  gpurun_out/pmcmem_<tag>.json (per-kernel averages per launch)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/pmcmem_$TAG; rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-c3 --sync-steps $*"
# one small group per pass (a pass that asks one block for more counters than it has aborts rocprofv3 and then hangs in its
# signal handler: every pass runs under `timeout`)
pass() { timeout 150 rocprofv3 --pmc "${@:2}" --output-format csv -d $OUT/$1 -o $1 -- python3 bench.py $ARGS > /dev/null 2> $OUT/$1.err || echo "pass $1 failed / timed out"; }
pass a TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass a2 TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_WRITE_WAVEFRONTS_sum
pass b TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
pass b2 TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass b3 TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_READ_sum
pass c TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
pass c2 TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
pass d SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INST_LEVEL_LDS
pass e TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass e2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
python3 - "$OUT" "$TAG" <<'PY'
import sys, json, re
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
from pmc_aggregate import aggregate
from tracerboy_amd import build as tb_build
out, tag = sys.argv[1], sys.argv[2]
def key(k):
    m = re.search(r"(pt_persistent<[^>]*>|pt_primary<[^>]*>|pt_split<[^>]*>|accumulate_samples_kernel)", k)
    return m.group(1) if m else None
res = aggregate(out + "/**/*counter_collection.csv", lambda k: key(k) is not None and "63u" not in key(k), key)   # real launches only (pmc_aggregate.py)
res["_kernel_digest"] = {"digest": tb_build.kernel_digest()}
json.dump(res, open("gpurun_out/pmcmem_%s.json" % tag, "w"), indent=1)
del res["_kernel_digest"]
for k, d in res.items():
    if d.get("SQ_WAVE_CYCLES", 0) < 1e6 and d.get("GRBM_GUI_ACTIVE", 0) < 1e7: continue
    print(k); print("  ", {c: (round(v / 1e6, 2) if v > 1000 else round(v, 3)) for c, v in sorted(d.items())})
PY
tail -1 $OUT/*.err | grep -i "error\|invalid\|unknown" | head
