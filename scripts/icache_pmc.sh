cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -i "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQC_TC_INST[A-Z_]*" | sort -u | tr '\n' ' '; echo
for tag in c4 c2; do
  if [ $tag = c4 ]; then ARGS="--scene proc1:700000 --builder 4 --width 3840 --height 2160 --spp 8 --depth 6 --legs none"; else ARGS="--legs none"; fi
  OUT=gpurun_out/ic_$tag; rm -rf $OUT
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT -o a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $ARGS > /dev/null 2> $OUT.err
  python3 - $OUT <<'PY'
import sys; sys.path.insert(0,"scripts")
from pmc_aggregate import aggregate, pt_key
res = aggregate(sys.argv[1] + "/**/*counter_collection.csv", lambda k: "pt_" in k and "63u" not in k, pt_key)
for k, d in res.items(): print(k[:70], {c: round(v/1e6,2) for c, v in sorted(d.items())})
PY
done
