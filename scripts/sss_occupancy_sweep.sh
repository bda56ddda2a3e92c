#!/bin/bash
# what do the register spills of the `sss` kernels cost?  For 3 / 4 / 5 / 6 waves per SIMD (libraries from
# scripts/build_sss_sweep.py) and the C4- / C5-class 4K scenes: Msamples/s, HBM bytes written per launch (WRITE_SIZE; the sample
# buffer is W x H x spp x 16 B of it, the rest is scratch), VALU lane utilisation, VALU and texture-addresser busy.
#   bash scripts/sss_occupancy_sweep.sh   ->  gpurun_out/sss_sweep.json
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/sss_sweep; rm -rf $OUT; mkdir -p $OUT
cp tracerboy_amd/libtracerboy_hip.so $OUT/lib_default.so
trap 'cp $OUT/lib_default.so tracerboy_amd/libtracerboy_hip.so' EXIT   # an interrupted run must not leave a sweep build in the tree
for W in 3 4 5 6; do
  cp tracerboy_amd/_sweep/libtracerboy_hip_sss$W.so tracerboy_amd/libtracerboy_hip.so
  for CFG in c4 c5; do
    if [ $CFG = c4 ]; then SC="--scene proc1:700000 --depth 6"; else SC="--scene proc2:2980000 --depth 16"; fi
    ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-c3 --no-readback --sync-steps --builder 4 --width 3840 --height 2160 --spp 8 $SC"
    timeout 200 python3 bench.py $ARGS > $OUT/${CFG}_w$W.bench.json 2> $OUT/${CFG}_w$W.bench.err
    pass() { timeout 200 rocprofv3 --pmc "${@:2}" --output-format csv -d $OUT/${CFG}_w$W/$1 -o $1 -- python3 bench.py $ARGS > /dev/null 2> $OUT/${CFG}_w$W.$1.err || echo "pass $1 failed"; }
    pass a WRITE_SIZE
    pass b SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
    pass c TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum GRBM_GUI_ACTIVE
  done
done
cp $OUT/lib_default.so tracerboy_amd/libtracerboy_hip.so
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, json, re
out = sys.argv[1]; res = {}
for cfg in ("c4", "c5"):
    for w in (3, 4, 5, 6):
        agg = collections.defaultdict(float); disp = collections.defaultdict(set)
        for f in glob.glob("%s/%s_w%d/**/*counter_collection.csv" % (out, cfg, w), recursive=True):
            for r in csv.DictReader(open(f)):
                if "pt_persistent" not in r["Kernel_Name"]: continue
                agg[(r["Kernel_Name"][:70], r["Counter_Name"])] += float(r["Counter_Value"]); disp[(r["Kernel_Name"][:70], r["Counter_Name"])].add(r["Dispatch_Id"])
        per = collections.defaultdict(dict)
        for (k, c), v in agg.items(): per[k][c] = v / max(1, len(disp[(k, c)]))
        main = max(per.items(), key=lambda kv: kv[1].get("SQ_WAVE_CYCLES", 0), default=(None, {}))
        d = main[1]
        try: bench = json.loads(open("%s/%s_w%d.bench.json" % (out, cfg, w)).read().strip().splitlines()[-1])
        except Exception: bench = {}
        cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8.0
        row = {"kernel": main[0], "Msamples_per_s": bench.get("value"), "avg_launch_ms": (bench.get("roofline") or {}).get("avg_launch_ms"),
               "write_GB_per_launch": round(d.get("WRITE_SIZE", 0) * 1024 / 1e9, 2), "sample_buffer_GB": round(3840 * 2160 * 8 * 16 / 1e9, 2),
               "valu_insts_G": round(d.get("SQ_INSTS_VALU", 0) / 1e9, 2), "vmem_rd_G": round(d.get("SQ_INSTS_VMEM_RD", 0) / 1e9, 3), "vmem_wr_G": round(d.get("SQ_INSTS_VMEM_WR", 0) / 1e9, 3),
               "lane_util": round(d["SQ_THREAD_CYCLES_VALU"] / (64 * d["SQ_ACTIVE_INST_VALU"]), 3) if d.get("SQ_ACTIVE_INST_VALU") else None,
               "valu_busy": round(4 * d.get("SQ_ACTIVE_INST_VALU", 0) / (1024 * cyc), 3) if cyc else None,
               "ta_busy": round(d.get("TA_TA_BUSY_sum", 0) / (256 * cyc), 3) if cyc else None,
               "wait_any": round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 3) if d.get("SQ_WAVE_CYCLES") else None}
        row["scratch_write_GB"] = round(row["write_GB_per_launch"] - row["sample_buffer_GB"], 2)
        res["%s_w%d" % (cfg, w)] = row
        print(cfg, w, row)
json.dump(res, open("gpurun_out/sss_sweep.json", "w"), indent=1)
PY
