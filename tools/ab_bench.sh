#!/bin/bash
# Same-box A/B of two builds of the library on the whole step: tools/ab_bench.sh <libA.so> <libB.so> [rounds]
# (boxes of the pool differ by +-4 %: only numbers from ONE call compare). Alternates A, B, A, B...; prints ms_per_step of each run.
a=$1; b=$2; n=${3:-2}
for i in $(seq $n); do
  for lib in $a $b; do
    MPN_LIB=$lib python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'], d['value'])"
  done
done
