#!/bin/bash
# same-box A/B of two library builds on the dominant 3x3 shape: tools/ab_c3.sh <libA.so> <libB.so> [rounds]
cd "$(dirname "$0")/.."
a=$1; b=$2; n=${3:-2}
for i in $(seq $n); do
  for lib in $a $b; do MPN_LIB=$lib python tools/ko_c3.py 2>/dev/null | tail -1; done
done
