#!/bin/bash
# Same-box A/B of a bench.py flag on the whole step: tools/ab_flag.sh "<flag>" [rounds]  (alternates with / without the flag)
flag=$1; n=${2:-2}
for i in $(seq $n); do
  for f in "$flag" ""; do
    python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$f]', d['ms_per_step'], d['value'])"
  done
done
