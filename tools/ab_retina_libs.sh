#!/bin/bash
# same-box A/B of the detector head's step (BASELINE config 4) on several library builds: tools/ab_retina_libs.sh rounds lib1.so lib2.so ...
cd "$(dirname "$0")/.."
n=$1; shift
for i in $(seq $n); do
  for lib in "$@"; do
    MPN_LIB=$lib python - <<'PY' 2>/dev/null | grep -v amdgpu.ids
import os, json
from bench_legs import retinanet_benchmark
r = retinanet_benchmark(16)
print(os.environ["MPN_LIB"], "ms_per_step", r["ms_per_step"], "images/s", r["images_per_s"], "inference ms", r["inference_ms_per_batch"], "tower launch us", r["dominant_kernel"]["launch_us"])
PY
  done
done
