#!/bin/bash
# PMC pairs (lock-step kernel against the split-role kernel) for DESIGN.md's table "where the split-role kernel loses":
#   bash scripts/split_pmc.sh   -> profiles/r4/split_pmc_{c2,c3,c4}_{lockstep,split}.json (via gpurun_out/)
set -u
cd "$GRAFT_REPO_ROOT"
C3="--scene proc0:870000 --builder 4 --spp 32 --depth 6 --no-c3 --sync-steps --opt primary_prepass=0"
C4="--scene proc1:700000 --builder 4 --width 3840 --height 2160 --spp 8 --depth 6 --no-c3 --sync-steps --opt primary_prepass=0"
bash scripts/pmc_quick.sh c2_lockstep --no-c3 --sync-steps | tail -2
bash scripts/pmc_quick.sh c2_split --no-c3 --sync-steps --opt pipeline=4 --opt split_trav=6 --opt split_shade=10 | tail -2
bash scripts/pmc_quick.sh c3_lockstep $C3 | tail -2
bash scripts/pmc_quick.sh c3_split $C3 --opt pipeline=4 --opt split_trav=8 --opt split_shade=8 | tail -2
bash scripts/pmc_quick.sh c4_lockstep $C4 | tail -2
bash scripts/pmc_quick.sh c4_split $C4 --opt pipeline=4 --opt split_trav=6 --opt split_shade=10 | tail -2
