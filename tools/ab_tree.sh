#!/bin/bash
# Same-box A/B of two whole trees (library + host code): tools/ab_tree.sh <dirA> <dirB> [rounds]; prints ms_per_step of each run
a=$1; b=$2; n=${3:-2}
for i in $(seq $n); do
  for d in $a $b; do
    (cd $d && python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', d['ms_per_step'], d['value'])")
  done
done
