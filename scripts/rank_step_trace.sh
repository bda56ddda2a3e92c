#!/bin/bash
# Kernel trace of one GPU playing rank RANK of WORLD on a bench.py workload (scripts/rank_step.py): start / end of every kernel of the
# last burst relative to the first, per queue.    bash scripts/rank_step_trace.sh NAME LEG WORLD RANK [rank_step.py args]
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
NAME=$1; shift
OUT=gpurun_out/trace_$NAME; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 scripts/rank_step.py "$@" > $OUT/run.json 2> $OUT/err.txt
cat $OUT/run.json
python3 - "$OUT" "$NAME" <<'PY'
import csv, glob, sys
out, name = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        short = "fold" if "accumulate" in n else ("primary" if "pt_primary" in n else ("persistent" if "pt_persistent" in n else ("pack" if "pack_owned" in n else n[:40])))
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], short, int(r["Grid_Size_X"])))
rows.sort()
rows = rows[-60:]
t0 = rows[0][0]
with open("gpurun_out/r5/trace_%s.csv" % name, "w") as g:
    g.write("start_us,end_us,duration_us,queue,kernel,grid\n")
    for s, e, q, k, gx in rows: g.write("%.1f,%.1f,%.1f,%s,%s,%d\n" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, k, gx))
PY
tail -40 gpurun_out/r5/trace_$NAME.csv
