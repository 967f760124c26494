#!/bin/bash
# Where the persistent 3x3 kernel's time goes, by knock-out (same box, one call): tools/ko_c3.sh  (after building the variants:
#   for k in 1 2 3 4 8 16 24; do tools/build_variant.sh ko$k conv3x3.hip "-DMPN_KO=$k"; done)
# 1 = no epilogue, 2 = no halo staging in the tile loop, 3 = neither, 4 = halo loads without the commit, 8 = halo loads cache-hot,
# 16 = epilogue without its global stores, 24 = 8 + 16
cd "$(dirname "$0")/.."
python tools/ko_c3.py 2>/dev/null | tail -1
for k in 1 2 3 4 8 16 24; do MPN_LIB=multiposenet_amd/libmpn_hip_ko$k.so python tools/ko_c3.py 2>/dev/null | tail -1; done
python tools/ko_c3.py 2>/dev/null | tail -1
