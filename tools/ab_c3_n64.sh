#!/bin/bash
# round 6 A/B (same box, alternating): 64-channel tiles on the channel-split kernel with weights one / two stages ahead, multi-chunk tiles routed there
S="32 128 128 512 64  16 112 176 64 64  16 56 88 64 64"
for rep in 1 2; do
  for v in "" _bd2 _all64 _all64bd2; do
    lib=multiposenet_amd/libmpn_hip$v.so
    echo "== $lib (rep $rep)"
    MPN_LIB=$lib python tools/time_c3.py $S 2>/dev/null
  done
done
