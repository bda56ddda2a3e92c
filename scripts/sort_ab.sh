#!/bin/bash
# Material-sorted shading on / off in the wavefront pipeline (option wavefront_sort): Msamples/s and, per kernel, VALU lane utilisation
# (SQ_THREAD_CYCLES_VALU / 64 / SQ_ACTIVE_INST_VALU) from a PMC pass.   bash scripts/sort_ab.sh <tag> <bench args...>
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=$1; shift
for SORT in 0 1; do
  python3 bench.py --no-c3 --no-cpu-baseline --pipeline 2 --opt wavefront_sort=$SORT --steps 3 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$TAG sort=$SORT', d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step', d['config']['kernel_variant'])"
  OUT=gpurun_out/sort_${TAG}_$SORT; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT -o p -- python3 bench.py --no-c3 --no-cpu-baseline --pipeline 2 --opt wavefront_sort=$SORT --steps 1 --warmup 0 "$@" > /dev/null 2> $OUT/err.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python3 bench.py --no-c3 --no-cpu-baseline --pipeline 2 --opt wavefront_sort=$SORT --steps 1 --warmup 0 "$@" > /dev/null 2>> $OUT/err.txt
  python3 - "$OUT" "$TAG" "$SORT" <<'PY'
import csv, glob, collections, sys, json, re
out, tag, sort = sys.argv[1:4]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(wf_\w+|pt_\w+)", r["Kernel_Name"])
        if m: agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
tim = {}
for f in glob.glob(out + "/t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(wf_\w+|pt_\w+)", r["Name"])
        if m: tim[m.group(1)] = tim.get(m.group(1), 0) + float(r["TotalDurationNs"]) / 1e6
res = {}
for k, d in sorted(agg.items()):
    if d.get("SQ_ACTIVE_INST_VALU"):
        res[k] = {"lane_util": round(d["SQ_THREAD_CYCLES_VALU"] / 64 / d["SQ_ACTIVE_INST_VALU"], 3), "valu_insts_M": round(d["SQ_INSTS_VALU"] / 1e6, 1), "total_ms": round(tim.get(k, 0), 2)}
print(tag, "sort=" + sort, json.dumps(res))
json.dump(res, open("gpurun_out/sort_%s_%s.json" % (tag, sort), "w"), indent=1)
PY
done
