#!/bin/bash
# same-box A/B of the default bench step under an environment switch: tools/ab_env.sh VAR rounds [bench args]  (VAR=0 / VAR=1 alternating)
cd "$(dirname "$0")/.."
var=$1; n=$2; shift 2
for i in $(seq $n); do
  for v in 0 1; do
    env $var=$v python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', d['ms_per_step'], d['value'], d['config']['final_total_loss'])"
  done
done
