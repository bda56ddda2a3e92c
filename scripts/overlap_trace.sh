#!/bin/bash
# Kernel trace of the DEFAULT bench (launches of consecutive steps overlap on the context's two side streams): start / end of every
# path-tracing launch and sample fold, per queue, relative to the first -- the evidence that a step (18.5 ms) is shorter than a launch
# run alone (19.6 ms) because launch k + 1 starts inside launch k's tail.
#   bash scripts/overlap_trace.sh [name [bench args...]]   ->  gpurun_out/<name>.csv (+ a summary line); default name c2_overlap_trace, the default bench
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
NAME=${1:-c2_overlap_trace}; shift || true
OUT=gpurun_out/overlap_$NAME; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-c3 --no-readback "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 - "$OUT" "$NAME" <<'PY'
import csv, glob, sys
out, name = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "pt_persistent" in n or "accumulate_samples" in n or "pt_split" in n or "pt_primary" in n:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "queue %s stream %s" % (r["Queue_Id"], r.get("Stream_Id", "?")), "fold" if "accumulate" in n else ("launch" if int(r["Grid_Size_X"]) > int(r["Workgroup_Size_X"]) else "warm"), n[n.find("pt_"):][:60] if "pt_" in n else "accumulate_samples_kernel"))
rows.sort()
t0 = rows[0][0]
with open("gpurun_out/%s.csv" % name, "w") as g:
    g.write("start_us,end_us,duration_us,queue,kind,kernel\n")
    for s, e, q, k, n in rows: g.write("%.1f,%.1f,%.1f,%s,%s,\"%s\"\n" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, k, n))
L = [r for r in rows if r[3] == "launch" and (r[1] - r[0]) > 5e6]     # the timed launches (> 5 ms)
ov = [(L[i][1] - L[i + 1][0]) / 1e3 for i in range(len(L) - 1)]
span = (L[-1][1] - L[0][0]) / 1e6 / max(len(L), 1)
print({"launches": len(L), "queues": sorted({r[2] for r in L}), "mean_launch_ms": round(sum(r[1] - r[0] for r in L) / 1e6 / len(L), 3), "ms_per_step_from_trace": round((L[-1][1] - L[0][0]) / 1e6 / (len(L) - 1), 3) if len(L) > 1 else None,
       "overlap_of_consecutive_launches_us": [round(x, 1) for x in ov[-8:]]})
PY
