cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/r6/gsweep; rm -rf $OUT; mkdir -p $OUT
for g in 2 8 32; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/g$g -o p -- python3 scripts/mix_step.py c2 frame_group=$g > /dev/null 2> $OUT/g$g.err
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/h$g -o p -- python3 scripts/mix_step.py c2 frame_group=$g > /dev/null 2> $OUT/h$g.err
done
python3 - <<'PY'
import sys; sys.path.insert(0, "scripts")
from pmc_aggregate import aggregate
for g in (2, 8, 32):
    row = {}
    for t in ("g", "h"):
        agg = aggregate("gpurun_out/r6/gsweep/%s%d/**/*counter_collection.csv" % (t, g), lambda k: "pt_persistent<" in k)
        name, c = max(agg.items(), key=lambda kv: kv[1].get("SQ_INSTS_VALU", kv[1].get("SQ_WAIT_ANY", 0)))
        row.update({k: round(v / 1e6, 1) for k, v in c.items() if k.startswith(("SQ_", "GRBM"))})
    print(g, row)
PY
TB_LIB=$PWD/tracerboy_amd/_sweep/libtracerboy_hip_d3.so python3 scripts/mix_step.py c2 --time
rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/d3 -o p -- python3 scripts/mix_step.py c2 > /dev/null 2>&1
