#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
for spp in 8 32; do for world in 1 8; do python3 scripts/rank_step.py c4 $world 0 --spp $spp 2>/dev/null; done; done
python3 scripts/rank_step.py c4 8 0 --opt primary_prepass=0 2>/dev/null
python3 scripts/rank_step.py c4 8 0 --opt frame_group=1 2>/dev/null
python3 scripts/rank_step.py c4 8 0 --opt frame_group=4 2>/dev/null
python3 scripts/rank_step.py c4 8 0 --opt frame_group=8 2>/dev/null
python3 scripts/rank_step.py c4 8 0 --opt overlap_launches=0 2>/dev/null
python3 scripts/rank_step.py vwvan 8 0 2>/dev/null
python3 scripts/rank_step.py vwvan 8 5 2>/dev/null
python3 scripts/rank_step.py vwvan 8 0 --spp 32 2>/dev/null
python3 scripts/rank_step.py vwvan 1 0 --spp 32 2>/dev/null
bash scripts/rank_step_trace.sh c4_w8 c4 8 0
