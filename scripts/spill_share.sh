#!/bin/bash
# What share of the timed kernel's vector-memory instructions moves spilled registers?  There is no counter for scratch accesses
# (scratch_load / scratch_store are FLAT-encoded like global ones), so the same kernels are built a second time WITHOUT their
# occupancy bound (-DTB_NO_OCCUPANCY_BOUND: all the registers they want, no spills; the host's plan -- copy, split stack, pre-pass -- is
# unchanged, so the control flow and every real load and store are the same) and the two builds' SQ_INSTS_VMEM_RD / _WR per launch
# are subtracted.  Here (CPU):  python scripts/build_variant.py nobound --flags=-DTB_NO_OCCUPANCY_BOUND=1 --tus kernels/pt_variant_sss4.hip \
#                                 kernels/pt_variant_vol4.hip kernels/pt_variant_surf.hip
# GPU box:     bash scripts/spill_share.sh   ->  gpurun_out/r6/<leg>_spill_share.json (copy to profiles/rN/)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/r6/spill_share; rm -rf $OUT; mkdir -p $OUT
LEGS=${LEGS:-"c4 c5 vwvan vwvan_2level teapot"}
for tag in base nobound; do
  if [ "$tag" = base ]; then unset TB_LIB; else export TB_LIB=$PWD/tracerboy_amd/_sweep/libtracerboy_hip_$tag.so; fi
  for leg in $LEGS; do
    python3 scripts/rank_step.py $leg 1 0 2>/dev/null | sed "s/^/$tag /" >> $OUT/times.txt
    rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/pmc_${tag}_$leg -o p -- python3 scripts/rank_step.py $leg 1 0 --steps 2 > /dev/null 2> $OUT/pmc_${tag}_$leg.err
  done
done
unset TB_LIB
python3 - "$OUT" "$LEGS" <<'PY'
import sys, json
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
from pmc_aggregate import aggregate, pt_key
from tracerboy_amd import build as tb_build
out, legs = sys.argv[1], sys.argv[2].split()
times = {}
for l in open(out + "/times.txt"):
    tag, js = l.split(" ", 1); d = json.loads(js); times[(tag, d["leg"])] = min(d["ms_per_step"])
for leg in legs:
    rows = {}
    for tag in ("base", "nobound"):
        agg = aggregate("%s/pmc_%s_%s/**/*counter_collection.csv" % (out, tag, leg), lambda k: ("pt_persistent" in k or "pt_primary" in k) and "63u" not in k, pt_key)
        tot = {"SQ_INSTS_VMEM_RD": 0.0, "SQ_INSTS_VMEM_WR": 0.0, "SQ_INSTS_VALU": 0.0}
        for k, v in agg.items():
            for c in tot: tot[c] += v.get(c, 0.0)          # per launch: pre-pass + lock-step kernel
        rows[tag] = {**{c: int(v) for c, v in tot.items()}, "kernels": sorted(agg), "ms_per_step": times.get((tag, leg))}
    b, n = rows["base"], rows["nobound"]
    vm_b, vm_n = b["SQ_INSTS_VMEM_RD"] + b["SQ_INSTS_VMEM_WR"], n["SQ_INSTS_VMEM_RD"] + n["SQ_INSTS_VMEM_WR"]
    doc = {"workload": leg, "shipped": b, "no_occupancy_bound": n,
           "spill_loads_per_launch": b["SQ_INSTS_VMEM_RD"] - n["SQ_INSTS_VMEM_RD"], "spill_stores_per_launch": b["SQ_INSTS_VMEM_WR"] - n["SQ_INSTS_VMEM_WR"],
           "vmem_spill_share": round(1.0 - vm_n / vm_b, 4) if vm_b else None, "_kernel_digest": tb_build.kernel_digest(),
           "method": "SQ_INSTS_VMEM_RD + _WR per launch (pre-pass + lock-step kernel) of the shipped build minus the same kernels built without their occupancy bound (no spills; same plan, same control flow)"}
    json.dump(doc, open("gpurun_out/r6/%s_spill_share.json" % leg, "w"), indent=1)
    print(leg, doc["vmem_spill_share"], doc["spill_loads_per_launch"], doc["spill_stores_per_launch"], b["ms_per_step"], n["ms_per_step"])
PY
