#!/bin/bash
# prints "label value ms_per_step" for a list of bench.py argument strings (one per line on stdin); run on the GPU box
while IFS= read -r args; do
  [ -z "$args" ] && continue
  python bench.py --no-cpu-baseline --steps 5 --warmup 2 $args 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-70s %9.1f Msamples/s  %8.3f ms  boxes %.2f' % (sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['boxes_per_sample']))" "$args"
done
