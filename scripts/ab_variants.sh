#!/bin/bash
# A/B of experimental library builds (scripts/build_variant.py -> tracerboy_amd/_sweep/libtracerboy_hip_<tag>.so) against the tree's own:
# ms per step of the given legs (scripts/rank_step.py, async steps as bench.py times them) and, for PMC_LEGS, the per-launch
# vector-memory instruction counts of the lock-step kernel (SQ_INSTS_VMEM_WR ~ spill stores: the sample buffer and the split stack's
# overflow pushes are ~1 % of them).   LEGS="c4 c5 vwvan teapot" PMC_LEGS="c4 vwvan" bash scripts/ab_variants.sh NAME base tag1 tag2 ...
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
NAME=$1; shift
OUT=gpurun_out/r5/ab_$NAME; rm -rf $OUT; mkdir -p $OUT
LEGS=${LEGS:-"c4 c5 vwvan teapot"}; PMC_LEGS=${PMC_LEGS:-"c4"}
for tag in "$@"; do
  if [ "$tag" = base ]; then unset TB_LIB; else export TB_LIB=$PWD/tracerboy_amd/_sweep/libtracerboy_hip_$tag.so; fi
  for leg in $LEGS; do python3 scripts/rank_step.py $leg 1 0 2>/dev/null | sed "s/^/$tag /" | tee -a $OUT/times.txt | cut -c1-170; done
  for leg in $PMC_LEGS; do
    rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc_${tag}_$leg -o p -- python3 scripts/rank_step.py $leg 1 0 --steps 2 > /dev/null 2> $OUT/pmc_${tag}_$leg.err
  done
done
python3 - "$OUT" "$NAME" "$PMC_LEGS" "$@" <<'PY'
import sys, json, re
sys.path.insert(0, "scripts")
from pmc_aggregate import aggregate, pt_key
out, name, legs, tags = sys.argv[1], sys.argv[2], sys.argv[3].split(), sys.argv[4:]
res = {"times": [json.loads(l.split(" ", 1)[1]) | {"lib": l.split(" ", 1)[0]} for l in open(out + "/times.txt")], "pmc": {}}
for tag in tags:
    for leg in legs:
        agg = aggregate("%s/pmc_%s_%s/**/*counter_collection.csv" % (out, tag, leg), lambda k: "pt_persistent" in k and "63u" not in k, pt_key)
        big = max(agg.items(), key=lambda kv: kv[1].get("SQ_INSTS_VALU", 0)) if agg else (None, {})
        res["pmc"]["%s %s" % (tag, leg)] = {"kernel": big[0], **{k: round(v / 1e6, 2) for k, v in big[1].items() if k.startswith("SQ_")}, "launches": big[1].get("dispatches")}
        print(tag, leg, res["pmc"]["%s %s" % (tag, leg)])
json.dump(res, open("gpurun_out/r5/ab_%s.json" % name, "w"), indent=1)
PY
