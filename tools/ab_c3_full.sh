#!/bin/bash
# same-box A/B of two library builds over the 3x3 layer table and the fused data gradient: tools/ab_c3_full.sh <libA.so> <libB.so> [rounds]
cd "$(dirname "$0")/.."
a=$1; b=$2; n=${3:-2}
for i in $(seq $n); do
  for lib in $a $b; do
    echo "== $lib"
    MPN_LIB=$lib python tools/time_c3.py 2>/dev/null | grep -v amdgpu.ids
    MPN_LIB=$lib python tools/time_c3_bnr.py 2>/dev/null | grep -v amdgpu.ids
  done
done
