#!/bin/bash
# The round's committed counters, all of one build: kernel stats + PMC passes (profile_bench.sh) and memory-pipeline counters (pmc_mem.sh)
# of every workload bench.py reports: C2 (cornell-box 1080p x 64), C3 (870 k, 1080p x 128), C4- / C5-class 4K x 8, Teapot 1080p x 16,
# the reference's vw-van at 4K x 8 flattened and two-level -- with the trees bench.py's WORKLOADS build for them (builder 1, reinsertion passes per workload).
#   bash scripts/profile_all.sh [tags...]   -> gpurun_out/<tag>/..., gpurun_out/pmcmem_<tag>.json   (copy into profiles/rN/ afterwards: scripts/collect_profiles.py)
set -u
cd "$GRAFT_REPO_ROOT"
declare -A A
A[c2]=""
A[c3]="--scene proc0:870000 --builder 1 --opt reinsertion_passes=3 --opt reinsertion_share=3 --spp 128 --depth 6"
A[c4]="--scene proc1:700000 --builder 1 --opt reinsertion_passes=3 --opt reinsertion_share=3 --width 3840 --height 2160 --spp 8 --depth 6"
A[c5]="--scene proc2:2980000 --builder 1 --opt reinsertion_passes=1 --opt reinsertion_share=3 --width 3840 --height 2160 --spp 8 --depth 16"
A[teapot]="--scene tests/golden/scenes/Teapot/scene.pbrt --builder 1 --opt reinsertion_passes=0 --spp 16 --depth 8"
A[vwvan]="--scene tests/golden/scenes/vw-van/vw-van.pbrt --builder 1 --opt reinsertion_passes=3 --opt reinsertion_share=3 --width 3840 --height 2160 --spp 8 --depth 6"
A[vwvan_2level]="--scene tests/golden/scenes/vw-van/vw-van.pbrt --builder 1 --opt reinsertion_passes=3 --opt reinsertion_share=3 --width 3840 --height 2160 --spp 8 --depth 6 --opt flatten_instances=0"
for t in ${@:-c2 c3 c4 c5 teapot vwvan vwvan_2level}; do
  echo "== $t"
  TAG=$t BENCH_ARGS="${A[$t]}" bash scripts/profile_bench.sh > gpurun_out/profile_$t.log 2>&1; tail -4 gpurun_out/profile_$t.log | cut -c1-400
  bash scripts/pmc_mem.sh $t ${A[$t]} > gpurun_out/pmcmem_$t.log 2>&1; tail -3 gpurun_out/pmcmem_$t.log | cut -c1-400
done
