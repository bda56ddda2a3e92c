#!/bin/bash
# The round's committed counters and bench lines, all of the final build: profile_all.sh (kernel stats, PMC, memory counters of the seven workloads),
# spill_share.sh, the per-rank sweeps the scale legs' expected speed-ups are read from, the stress of the item lists, the GPU suite, and -- with the
# fresh counters copied into the box's profiles/ the way scripts/collect_profiles.py does here afterwards -- the default bench line.
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r6; mkdir -p $O
bash scripts/profile_all.sh 2>&1 | grep "^==\|Msamples" | cut -c1-200
bash scripts/spill_share.sh > $O/spill_share.log 2>&1; tail -6 $O/spill_share.log | cut -c1-300
python3 scripts/rank_imbalance.py $O/rank_imbalance.json > $O/rank_imbalance.log 2>&1; grep "world 8" $O/rank_imbalance.log
python3 scripts/rank_imbalance.py $O/rank_imbalance_32spp.json --spp 32 --steps 4 > $O/rank_imbalance_32spp.log 2>&1; grep "world 8" $O/rank_imbalance_32spp.log
timeout 900 python3 scripts/costly_first_stress.py 300 > $O/costly_first_stress.txt 2>&1; tail -1 $O/costly_first_stress.txt
timeout 1500 python3 -m pytest tests -m gpu -q > $O/gpu_suite.log 2>&1; tail -1 $O/gpu_suite.log
python3 scripts/collect_profiles.py r6 > /dev/null; cp $O/*_spill_share.json $O/rank_imbalance.json $O/rank_imbalance_32spp.json profiles/r6/
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-200 $O/bench_default.json
python3 bench.py --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2>/dev/null; cut -c1-200 $O/bench_steps20_warmup5.json
