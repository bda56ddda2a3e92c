#!/bin/bash
# A/B of one option with three PMC passes per side (issue mix, texture addresser, memory instructions), one launch at a time:
#   bash scripts/pmc_ab.sh <tag> <option>=<a>,<b> <bench args...>   ->  gpurun_out/pmcab_<tag>.json
# e.g.  bash scripts/pmc_ab.sh c3 node_layout=0,1 --scene proc0:870000 --spp 128 --depth 6
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=$1; OPT=${2%%=*}; VALS=${2#*=}; shift 2
OUT=gpurun_out/pmcab_$TAG; rm -rf $OUT; mkdir -p $OUT
for V in ${VALS//,/ }; do
  ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-c3 --no-readback --sync-steps --opt $OPT=$V $*"
  pass() { timeout 200 rocprofv3 --pmc "${@:2}" --output-format csv -d $OUT/v$V/$1 -o $1 -- python3 bench.py $ARGS > /dev/null 2> $OUT/v$V.$1.err || echo "pass $1 failed / timed out"; }
  python3 bench.py $ARGS > $OUT/v$V.bench.json 2> $OUT/v$V.bench.err
  pass a SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
  pass b TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
  pass c SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM
  pass d FETCH_SIZE
  pass e WRITE_SIZE
  pass f TCC_HIT_sum TCC_MISS_sum
done
python3 - "$OUT" "$TAG" "$VALS" <<'PY'
import csv, glob, collections, sys, json, re
out, tag, vals = sys.argv[1], sys.argv[2], sys.argv[3].split(",")
res = {}
for v in vals:
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for f in glob.glob("%s/v%s/**/*counter_collection.csv" % (out, v), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(pt_persistent<[^>]*>)", r["Kernel_Name"])
            if not m: continue
            agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"]); disp[(m.group(1), r["Counter_Name"])].add(r["Dispatch_Id"])
    per = {k: {c: x / max(1, len(disp[(k, c)])) for c, x in d.items()} for k, d in agg.items()}
    per = {k: d for k, d in per.items() if d.get("SQ_WAVE_CYCLES", 0) > 1e8}   # the timed launches, not the warm-up twins
    for k, d in per.items():
        cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8.0   # per XCD
        if d.get("SQ_ACTIVE_INST_VALU"): d["valu_lane_utilisation"] = d["SQ_THREAD_CYCLES_VALU"] / (d["SQ_ACTIVE_INST_VALU"] * 64)
        if cyc:
            d["ta_busy"] = d.get("TA_TA_BUSY_sum", 0) / (256 * cyc)
            d["valu_busy"] = d.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * cyc) if "SQ_ACTIVE_INST_VALU" in d else None
        if "FETCH_SIZE" in d: d["hbm_bytes"] = (2 * d["FETCH_SIZE"] + d.get("WRITE_SIZE", 0)) * 1024
    try: bench = json.loads(open("%s/v%s.bench.json" % (out, v)).read().strip().splitlines()[-1])
    except Exception: bench = {}
    res[v] = {"kernels": per, "Msamples_per_s": bench.get("value"), "ms_per_step": bench.get("ms_per_step"), "avg_launch_ms": (bench.get("roofline") or {}).get("avg_launch_ms")}
json.dump(res, open("gpurun_out/pmcab_%s.json" % tag, "w"), indent=1)
for v, r in res.items():
    print("==", v, r["Msamples_per_s"], r["ms_per_step"], r["avg_launch_ms"])
    for k, d in r["kernels"].items(): print("  ", k, {c: (round(x / 1e6, 1) if x > 1000 else round(x, 3)) for c, x in sorted(d.items()) if x is not None})
PY
