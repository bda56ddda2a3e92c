cd "$GRAFT_REPO_ROOT"
for leg in c2 c3 c4 teapot vwvan; do
  for o in "park_min=8" "park_min=4" "park_min=12" "park_min=16" "park_min=24" "park_min=32" "banded_items=1"; do
    echo -n "$leg $o: "; python3 scripts/async_rate.py $leg --reps 2 --steps 8 --opt $o 2>/dev/null | tail -1 | cut -c60-135
  done
done
