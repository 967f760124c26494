#!/bin/bash
# same-box comparison of several library builds over the 3x3 layer table and the fused data gradient: tools/ab_c3_libs.sh rounds lib1.so lib2.so ...
cd "$(dirname "$0")/.."
n=$1; shift
for i in $(seq $n); do
  for lib in "$@"; do
    echo "== $lib"
    MPN_LIB=$lib python tools/time_c3.py 2>/dev/null | grep -v amdgpu.ids
    MPN_LIB=$lib python tools/time_c3_bnr.py 2>/dev/null | grep -v amdgpu.ids
  done
done
