#!/bin/bash
# same-box A/B of the detector head's step (BASELINE config 4) with the towers' first layers merged / separate: tools/ab_retina.sh [rounds]
cd "$(dirname "$0")/.."
n=${1:-2}
for i in $(seq $n); do
  for m in 0 1; do
    MPN_RETINA_MERGE=$m python - <<'PY' 2>/dev/null | grep -v amdgpu.ids
import os, json
from bench_legs import retinanet_benchmark
r = retinanet_benchmark(16)
print("MPN_RETINA_MERGE=%s" % os.environ["MPN_RETINA_MERGE"], "ms_per_step", r["ms_per_step"], "images/s", r["images_per_s"], "inference ms", r["inference_ms_per_batch"], "losses", json.dumps(r["losses"]))
PY
  done
done
