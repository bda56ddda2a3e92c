#!/bin/bash
# The measurements docs/experiments/r6.md sections 2, 3 and 5 quote, regenerated in one go on the GPU box (about 3 minutes):
#   gpurun_out/r6/evidence/{async_rate.jsonl, guided_sync_ab.jsonl, wg_timeline_<case>.json}
# Needs tracerboy_amd/_sweep/libtracerboy_hip_tl.so for the timelines (python scripts/build_variant.py tl --flags=-DTB_WG_TIMELINE --tus
# kernels/pt_variant_matte5.hip kernels/pt_variant_sss4.hip kernels/pt_variant_env5.hip kernels/pt_variant_vol4.hip); skipped without it.
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/r6/evidence; rm -rf $OUT; mkdir -p $OUT
for a in "c2 --spp 16,32,64,128" "c4 --spp 8,32" "c4 --world 8 --spp 8,32" "vwvan --world 8 --spp 8,32" "c2 --opt guided_groups=2" "c4 --opt guided_groups=2"; do
  python3 scripts/async_rate.py $a 2>/dev/null | grep "^{" >> $OUT/async_rate.jsonl
done
for leg in c2 c3 c4 c5 teapot vwvan vwvan_2level; do for g in 0 1; do
  python3 scripts/mix_step.py $leg --time guided_groups=$g 2>/dev/null | grep "^{" | sed "s/^{/{\"guided_groups\": $g, /" >> $OUT/guided_sync_ab.jsonl
done; done
if [ -f tracerboy_amd/_sweep/libtracerboy_hip_tl.so ]; then
  export TB_LIB=$PWD/tracerboy_amd/_sweep/libtracerboy_hip_tl.so
  for a in "c2" "c2 --opt guided_groups=0" "c2 8 0" "c3" "c4" "c4 8 0" "c5 8 0" "vwvan 8 0" "c2 --opt frame_group=8 --opt guided_groups=0" "c2 --opt frame_group=2 --opt guided_groups=0"; do
    python3 scripts/wg_timeline.py $a --out "$OUT/wg_timeline_$(echo $a | tr -d '-' | tr ' =' '__').json" > /dev/null 2>&1
  done
fi
ls $OUT; cat $OUT/async_rate.jsonl | cut -c1-200
