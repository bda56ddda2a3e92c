#!/bin/bash
# The round's committed counters and bench lines, all of the final build: profile_all.sh (kernel stats, PMC, memory counters of the seven workloads),
# spill_share.sh, the per-rank sweeps the scale legs' expected speed-ups are read from, the default bench line.  Copy with scripts/collect_profiles.py r6.
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r6; mkdir -p $O
bash scripts/profile_all.sh 2>&1 | tail -30
bash scripts/spill_share.sh > $O/spill_share.log 2>&1; tail -6 $O/spill_share.log | cut -c1-300
python3 scripts/rank_imbalance.py $O/rank_imbalance.json > $O/rank_imbalance.log 2>&1; grep "world 8" $O/rank_imbalance.log
python3 scripts/rank_imbalance.py $O/rank_imbalance_32spp.json --spp 32 --steps 4 --worlds 1,8 > $O/rank_imbalance_32spp.log 2>&1; grep "world 8" $O/rank_imbalance_32spp.log
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json
