#!/bin/bash
# Round-5 final artifacts, two steps:
#   on the GPU box (gpurun):  tools/final_artifacts_r05.sh run      -> gpurun_out/bench_r05_final.json, prof_r05f/, r05f leg traces
#   afterwards, locally:      tools/final_artifacts_r05.sh collect  -> profiles/r05_*
if [ "$1" = "run" ]; then
  root=${GRAFT_REPO_ROOT:-/root/repo}
  cd $root && python bench.py > gpurun_out/bench_r05_final.json 2> gpurun_out/bench_r05_final.err || exit 1
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_r05f -o run -- python3 $root/bench.py --no-cpu-baseline > $root/gpurun_out/prof_r05f.log 2>&1 || exit 1
  cd $root && tools/profile_legs.sh r05f > gpurun_out/r05f_legs.txt 2>&1
  exit 0
fi
f=$(find gpurun_out/prof_r05f -name run_kernel_trace.csv | head -1)
python tools/trace_summary.py $f > /tmp/sum.txt
python tools/dominant_from_trace.py $f >> /tmp/sum.txt
python - <<'PY' >> /tmp/sum.txt
import json, re
d = json.loads(open('gpurun_out/bench_r05_final.json').read().strip().splitlines()[-1])
u = json.loads(re.search(r'\{"metric".*\}', open('gpurun_out/prof_r05f.log').read()).group(0))
print(f"(rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline; the bench line printed under "
      f"the profiler in that run: ms_per_step {u['ms_per_step']}, roofline.launch_us {u['roofline']['launch_us']}; the unprofiled default "
      f"run before it on the same box: ms_per_step {d['ms_per_step']}, value {d['value']}, roofline.frac {d['roofline']['frac']}, "
      f"launch_us {d['roofline']['launch_us']}; tools/trace_summary.py + tools/dominant_from_trace.py on the kernel trace)")
PY
cp /tmp/sum.txt profiles/r05_final_step_summary.txt
cp $(dirname $f)/run_kernel_stats.csv profiles/r05_final_kernel_stats.csv
cp gpurun_out/bench_r05_final.json profiles/r05_bench_default.json
tools/profile_legs.sh --collect r05f
cat profiles/r05_final_step_summary.txt
