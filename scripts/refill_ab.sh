#!/bin/bash
# the wavefront pipeline (pipeline 2) with wf_extend / wf_connect in their persistent form with in-kernel refill
# (option wavefront_refill = n) against the plain form and against the megakernel (pipeline 0).
#   bash scripts/refill_ab.sh   ->  gpurun_out/refill_ab.json  (Msamples/s per setting + per-kernel lane utilisation and time)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/refill_ab; rm -rf $OUT; mkdir -p $OUT
run() { # tag, bench args...
  TAG=$1; shift
  timeout 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-c3 --no-readback "$@" > $OUT/$TAG.bench.json 2> $OUT/$TAG.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$TAG.trace -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-c3 --no-readback "$@" > /dev/null 2>> $OUT/$TAG.err
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/$TAG.pmc -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-c3 --no-readback "$@" > /dev/null 2>> $OUT/$TAG.err
}
C3="--scene proc0:870000 --spp 32 --depth 6"
C5="--scene proc2:2980000 --builder 4 --width 3840 --height 2160 --spp 4 --depth 16"
run c3_mega $C3
run c5_mega $C5
for R in 0 16 32 48; do
  run c3_wf_r$R $C3 --pipeline 2 --opt wavefront_refill=$R
  run c5_wf_r$R $C5 --pipeline 2 --opt wavefront_refill=$R
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, json, re, os
out = sys.argv[1]; res = {}
for b in sorted(glob.glob(out + "/*.bench.json")):
    tag = os.path.basename(b)[:-len(".bench.json")]
    try: bench = json.loads(open(b).read().strip().splitlines()[-1])
    except Exception: bench = {}
    row = {"Msamples_per_s": bench.get("value"), "ms_per_step": bench.get("ms_per_step"), "kernels": {}}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("%s/%s.pmc/**/*counter_collection.csv" % (out, tag), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(wf_\w+|pt_persistent|accumulate_samples_kernel)", r["Kernel_Name"])
            if m: agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
    times = collections.defaultdict(float); total = 0.0
    for f in glob.glob("%s/%s.trace/**/*kernel_stats.csv" % (out, tag), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(wf_\w+|pt_persistent|accumulate_samples_kernel)", r["Name"])
            if m: times[m.group(1)] += float(r["TotalDurationNs"]); total += float(r["TotalDurationNs"])
    for k, d in agg.items():
        row["kernels"][k] = {"lane_util": round(d["SQ_THREAD_CYCLES_VALU"] / (64 * d["SQ_ACTIVE_INST_VALU"]), 3) if d.get("SQ_ACTIVE_INST_VALU") else None,
                             "valu_insts_G": round(d.get("SQ_INSTS_VALU", 0) / 1e9, 2), "time_share": round(times.get(k, 0) / total, 3) if total else None}
    res[tag] = row
    print(tag, row["Msamples_per_s"], {k: (v["lane_util"], v["time_share"]) for k, v in row["kernels"].items()})
json.dump(res, open("gpurun_out/refill_ab.json", "w"), indent=1)
PY
