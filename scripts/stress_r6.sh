#!/bin/bash
# Stress record of round 6 (the slot entry's layout and the place of the bind trigger changed for every frame-group kernel; LDS-resident scenes got
# the GUIDED copy): scripts/guided_stress.py + round 5's configurations of scripts/lost_item_stress.py.   bash scripts/stress_r6.sh > gpurun_out/r6/stress_summary.txt
cd "$GRAFT_REPO_ROOT"
python3 -c "import sys; sys.path.insert(0, '.'); from tracerboy_amd import build as b; print('kernel digest of the build under test:', b.kernel_digest())"
echo "== shrinking groups, cornell-box (LDS), every render against the oracle"; timeout 1200 python3 scripts/guided_stress.py 150 2>&1 | grep "bad"
run() { echo "== $1"; shift; env "$@" 2>&1 | grep "bad\|option" | tr '\n' ' '; echo; }
run "default policy (pre-pass by rule / trial, overlap by rule / trial), 300 renders a scene" timeout 900 python3 scripts/lost_item_stress.py 300 1
run "pre-pass forced" timeout 900 python3 scripts/lost_item_stress.py 300 2
run "launches always overlap, three batches a render" TB_STRESS_OPTIONS="overlap_launches=2 pooled_samples=72000" timeout 900 python3 scripts/lost_item_stress.py 200 2
run "stack split at 6 entries (22+ in global memory)" TB_STRESS_OPTIONS="stack_lds_cap=6 stack_overflow_max=64" timeout 900 python3 scripts/lost_item_stress.py 200 2
run "16-B hit records with a 4-bit stamp" TB_STRESS_OPTIONS="compact_stamp_bits=4" timeout 900 python3 scripts/lost_item_stress.py 200 2
echo "== fresh processes (4 renders each, pre-pass forced)"; bash scripts/first_render_stress.sh 30 2 | tail -2
