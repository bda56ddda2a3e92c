#!/bin/bash
# Stress record (rounds 4 and 5): scripts/lost_item_stress.py (200 x 120 x 9, three random streams over two buffers, every render against the oracle)
# in the configurations that matter for this round's changes, and fresh processes.   bash scripts/stress_r5.sh > gpurun_out/r5/stress_summary.txt
cd "$GRAFT_REPO_ROOT"
python3 -c "import sys; sys.path.insert(0, '.'); from tracerboy_amd import build as b; print('kernel digest of the build under test:', b.kernel_digest())"
run() { echo "== $1"; shift; env "$@" 2>&1 | grep "bad\|option" | tr '\n' ' '; echo; }
run "default policy (pre-pass by rule / trial, overlap by rule / trial), 300 renders a scene" timeout 900 python scripts/lost_item_stress.py 300 1
run "pre-pass forced" timeout 900 python scripts/lost_item_stress.py 300 2
run "split-role kernel (pipeline 4) where the feature set has it" TB_STRESS_OPTIONS="pipeline=4" timeout 900 python scripts/lost_item_stress.py 300 1
run "split-role kernel, 2 x 6 waves, ready threshold 8, groups of 2 frames" TB_STRESS_OPTIONS="pipeline=4 split_trav=2 split_shade=6 split_ready=8 split_frame_group=2" timeout 900 python scripts/lost_item_stress.py 200 1
run "launches never overlap" TB_STRESS_OPTIONS="overlap_launches=0" timeout 900 python scripts/lost_item_stress.py 200 2
run "launches always overlap, three batches a render" TB_STRESS_OPTIONS="overlap_launches=2 pooled_samples=72000" timeout 900 python scripts/lost_item_stress.py 200 2
run "stack split at 6 entries (22+ in global memory)" TB_STRESS_OPTIONS="stack_lds_cap=6 stack_overflow_max=64" timeout 900 python scripts/lost_item_stress.py 200 2
run "a rank of 8 of a tile split is not exercised here: tests/test_gpu_parity.py test_tile_split_* do that" true
echo "== fresh processes (4 renders each, pre-pass forced)"; bash scripts/first_render_stress.sh 30 2 | tail -2
echo "== fresh processes, split-role kernel"; TB_STRESS_OPTIONS="pipeline=4" bash scripts/first_render_stress.sh 20 1 | tail -2
python - <<'PY'
import sys; sys.path.insert(0, ".")
from tracerboy_amd import api
tb = api.TracerBoy(0)
print("pre-pass records rejected in a fresh context after nothing:", tb.GetOption("debug_prepass_rejects"))
PY
