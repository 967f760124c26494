#!/bin/bash
# Host-side sanitizer pass over the launchers (ASan / UBSan on the C++ that does pointer, shape and workspace arithmetic
# before a launch). Every csrc/*.hip is rebuilt with -fsanitize=address,undefined on the HOST half only (-fno-gpu-sanitize: device
# sanitizers need XNACK, which this pool does not offer), linked into multiposenet_amd/libmpn_hip_asan.so, and the tests that call
# the launchers WITHOUT a GPU (argument validation, workspace / partial-row formulas at full size, ABI closure) run under it with the
# sanitizer runtime preloaded into the interpreter. Usage: tools/sanitize_host.sh [log file]   (CPU container; no GPU needed)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
log=${1:-$root/profiles/r05_host_sanitizers.txt}
b=$root/multiposenet_amd/csrc/build/asan
mkdir -p $b
rt=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
flags="-O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-gpu-rdc -munsafe-fp-atomics -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -fno-sanitize-recover=undefined -shared-libsan"
pids=""
for f in $root/multiposenet_amd/csrc/*.hip; do
  o=$b/$(basename ${f%.hip}).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ $root/multiposenet_amd/csrc/common.h -nt $o ] || [ $root/include/mpn.h -nt $o ]; then
    /opt/rocm/bin/hipcc $flags -c $f -o $o &
    pids="$pids $!"
    if [ $(echo $pids | wc -w) -ge 6 ]; then wait $pids; pids=""; fi
  fi
done
wait $pids
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fsanitize=address,undefined -shared-libsan -o $root/multiposenet_amd/libmpn_hip_asan.so $b/*.o
{
  echo "# host-side ASan + UBSan pass (tools/sanitize_host.sh), $(date -u +%Y-%m-%dT%H:%MZ): csrc/*.hip host halves built with"
  echo "# $flags"
  echo "# runtime: $rt (preloaded); tests: the no-GPU launcher tests"
  cd $root
  ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    LD_PRELOAD=$rt MPN_LIB=$root/multiposenet_amd/libmpn_hip_asan.so \
    python -m pytest tests/test_abi.py tests/test_host_sizes.py tests/test_host_logic.py -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15
} | tee $log
