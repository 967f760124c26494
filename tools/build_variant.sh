#!/bin/bash
# usage: tools/build_variant.sh NAME file.hip[,file2.hip...] "-DFLAG ..."  -> multiposenet_amd/libmpn_hip_NAME.so (same objects, the
# named files rebuilt with extra flags); run with MPN_LIB=multiposenet_amd/libmpn_hip_NAME.so for a same-box A/B of compile-time variants
set -e
name=$1; files=${2//,/ }; flags=$3
root=$(cd "$(dirname "$0")/.." && pwd)
b=$root/multiposenet_amd/csrc/build
mkdir -p $b/exp_$name
for f in $files; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-gpu-rdc -munsafe-fp-atomics $flags \
    -c $root/multiposenet_amd/csrc/$f -o $b/exp_$name/${f%.hip}.o &
done
wait
objs=""
for o in $b/*.o; do
  bn=$(basename $o)
  if [ -f "$b/exp_$name/$bn" ]; then objs="$objs $b/exp_$name/$bn"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/multiposenet_amd/libmpn_hip_$name.so $objs
echo built $root/multiposenet_amd/libmpn_hip_$name.so
