#!/bin/bash
# Like ab_flags.sh, but times the three big scenes with the pre-pass never / asked for (scripts/prepass_ab.py): bash scripts/ab_flags_prepass.sh "<flags 1>" ...
set -u
cd "$GRAFT_REPO_ROOT"
for FL in "$@"; do
  TB_EXTRA_FLAGS="$FL" timeout 600 python3 -m tracerboy_amd.build --force > /dev/null 2>&1 || { echo "build failed for [$FL]"; continue; }
  echo "[$FL]"; timeout 300 python3 scripts/prepass_ab.py --reps 3 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); print('   %-46s never %7.1f  asked for %7.1f   (layout C: %7.1f / %7.1f)' % (d['case'], d['layoutB_prepass0_Msamples_s'], d['layoutB_prepass1_Msamples_s'], d['layoutC_prepass0_Msamples_s'], d['layoutC_prepass1_Msamples_s']))"
done
TB_EXTRA_FLAGS="" python3 -m tracerboy_amd.build --force > /dev/null 2>&1
