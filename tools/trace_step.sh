#!/bin/bash
# kernel trace of the keypoint train step alone (no roofline / CPU legs): tools/trace_step.sh <tag>  -> gpurun_out/prof_<tag>/ and
# gpurun_out/<tag>_step_summary.txt (run on the GPU box; the program sits directly behind `--`)
tag=${1:-step}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/prof_$tag -o run -- python3 $root/bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 5 > $root/gpurun_out/prof_$tag.log 2>&1 || { tail -5 $root/gpurun_out/prof_$tag.log; exit 1; }
f=$(find $root/gpurun_out/prof_$tag -name 'run_kernel_trace.csv' | head -1)
python3 $root/tools/trace_summary.py $f > $root/gpurun_out/${tag}_step_summary.txt
cat $root/gpurun_out/${tag}_step_summary.txt
