#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + stats, then separate PMC passes
# (never --pmc together with trace domains).  Summaries are copied into profiles/ by hand afterwards.
#   TAG=r1 BENCH_ARGS="--scene proc0:870000 --spp 8" bash scripts/profile_r1.sh
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r1}
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
# --sync-steps: one launch at a time, so that a kernel's duration in the trace is its own (bench.py's default overlaps the launches of
# consecutive steps; its roofline block times non-overlapped launches the same way)
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --sync-steps ${BENCH_ARGS:-}"
python3 bench.py $ARGS > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -o sq -- python3 bench.py $ARGS > $OUT/bench_sq.json 2> $OUT/sq.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds -o lds -- python3 bench.py $ARGS > $OUT/bench_lds.json 2> $OUT/lds.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/pmc_tcc -o tcc -- python3 bench.py $ARGS > $OUT/bench_tcc.json 2> $OUT/tcc.err
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_util -o util -- python3 bench.py $ARGS > $OUT/bench_util.json 2> $OUT/util.err
find $OUT -name "*.csv" -size +0 | head -40
for f in $(find $OUT -name "*kernel_stats.csv"); do echo "== $f"; head -6 "$f"; done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, json
out = sys.argv[1]
summary = {}
for tag in ("fetch", "write", "sq", "lds", "tcc", "util"):
    for f in glob.glob("%s/pmc_%s/**/*counter_collection.csv" % (out, tag), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "pt_persistent" not in k and "accumulate_samples" not in k: continue
            agg[k[:80]][r["Counter_Name"]] += float(r["Counter_Value"])
            seen.add((k, r.get("Dispatch_Id")))
        for k, v in agg.items():
            d = len({s for s in seen if s[0][:80] == k})
            summary.setdefault(k, {})[tag] = {"dispatches": d, **{c: val / max(d, 1) for c, val in v.items()}}
print(json.dumps(summary, indent=1))
open(out + "/pmc_summary.json", "w").write(json.dumps(summary, indent=1))
PY
tail -3 $OUT/*.err | head -40
