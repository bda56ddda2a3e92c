#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + stats, then separate PMC passes of the SAME bench.py command
# (never --pmc together with trace domains).  Writes gpurun_out/<TAG>/{kernel_stats.csv, pmc_summary.json, bench.json}; the
# summaries are copied into profiles/rN/<TAG>_* by hand afterwards (bench.py reads profiles/rN/<TAG>_pmc_summary.json).
#   TAG=c2 bash scripts/profile_bench.sh
#   TAG=c3 BENCH_ARGS="--scene proc0:870000 --spp 128 --depth 6" bash scripts/profile_bench.sh
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=${TAG:-c2}
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
# --sync-steps: one launch at a time, so that a kernel's duration in the trace is its own (bench.py's default overlaps the launches of
# consecutive steps; its roofline block times non-overlapped launches the same way)
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-c3 --sync-steps ${BENCH_ARGS:-}"
python3 bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py $ARGS > /dev/null 2> $OUT/trace.err
pass() { rocprofv3 --pmc "${@:2}" --output-format csv -d $OUT/pmc_$1 -o $1 -- python3 bench.py $ARGS > /dev/null 2> $OUT/$1.err; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE
pass tcc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
pass util SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cp "$f" $OUT/kernel_stats_all_dispatches.csv; done
# what is committed as <TAG>_kernel_stats.csv: time per REAL launch (the zero-frame dispatches that warm the side streams dropped)
python3 scripts/kernel_stats_real.py $OUT/trace $OUT/kernel_stats.csv
python3 - "$OUT" <<'PY'
import sys, json
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
from pmc_aggregate import aggregate
from tracerboy_amd import build as tb_build
out = sys.argv[1]
summary = {}
keep = lambda k: "pt_persistent" in k or "accumulate_samples" in k or "pt_primary" in k or "pt_split" in k
for tag in ("fetch", "write", "sq", "lds", "tcc", "util"):
    # real launches only: the zero-frame dispatches the host warms its streams with are dropped before averaging (pmc_aggregate.py)
    for k, v in aggregate("%s/pmc_%s/**/*counter_collection.csv" % (out, tag), keep).items():
        summary.setdefault(k, {})[tag] = v
summary["_kernel_digest"] = tb_build.kernel_digest()   # bench.py compares it with the code it runs
open(out + "/pmc_summary.json", "w").write(json.dumps(summary, indent=1))
for k, v in summary.items():
    if k.startswith("_"): continue
    print(k, {t: {c: round(x / 1e6, 2) for c, x in d.items() if c not in ("dispatches", "warm_dispatches_dropped")} for t, d in v.items()})
PY
cat $OUT/bench.json
tail -2 $OUT/*.err | grep -v "^$" | head -30
