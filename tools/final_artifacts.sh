#!/bin/bash
# After a gpurun of:  python bench.py > gpurun_out/bench_r01_final.json ; rocprofv3 --kernel-trace --stats --output-format csv
#   -d gpurun_out/prof_r01f -o run -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_r01f.log
# copy the judged summaries into profiles/ (usage: tools/final_artifacts.sh [round tag, default r01])
tag=${1:-r01}
python tools/trace_summary.py gpurun_out/prof_${tag}f/run_kernel_trace.csv > /tmp/sum.txt
python tools/dominant_from_trace.py gpurun_out/prof_${tag}f/run_kernel_trace.csv >> /tmp/sum.txt
python - "$tag" <<'PY' >> /tmp/sum.txt
import json, re, sys
tag = sys.argv[1]
d = json.loads(open(f'gpurun_out/bench_{tag}_final.json').read().strip().splitlines()[-1])
u = json.loads(re.search(r'\{"metric".*\}', open(f'gpurun_out/prof_{tag}f.log').read()).group(0))
print(f"(rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline; the bench line printed under "
      f"the profiler in that run: ms_per_step {u['ms_per_step']}, roofline.launch_us {u['roofline']['launch_us']}; the unprofiled default "
      f"run before it on the same box: ms_per_step {d['ms_per_step']}, value {d['value']}, roofline.frac {d['roofline']['frac']}, "
      f"launch_us {d['roofline']['launch_us']}; tools/trace_summary.py + tools/dominant_from_trace.py on the kernel trace)")
PY
cp /tmp/sum.txt profiles/${tag}_final_step_summary.txt
cp gpurun_out/prof_${tag}f/run_kernel_stats.csv profiles/${tag}_final_kernel_stats.csv
cp gpurun_out/bench_${tag}_final.json profiles/${tag}_bench_default.json
