#!/bin/bash
# rocprofv3 kernel traces of the detector (cfg4) and PRN (cfg5) train steps -> gpurun_out/prof_<tag>_{detector,prn}/ and their
# step summaries gpurun_out/<tag>_<leg>_step_summary.txt: tools/profile_legs.sh <tag>   (run on the GPU box; the program sits
# directly behind `--`). gpurun merges only gpurun_out/ back: copy the summaries into profiles/ afterwards
# (tools/profile_legs.sh --collect <tag>).
if [ "$1" = "--collect" ]; then
  tag=$2
  for leg in detector prn; do
    cp gpurun_out/${tag}_${leg}_step_summary.txt profiles/${tag}_${leg}_step_summary.txt
    f=$(find gpurun_out/prof_${tag}_$leg -name 'run_kernel_stats.csv' | head -1)
    [ -n "$f" ] && cp $f profiles/${tag}_${leg}_kernel_stats.csv
  done
  exit 0
fi
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for leg in detector prn; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_${tag}_$leg -o run -- python3 $root/tools/leg_step.py $leg 5 > $root/gpurun_out/prof_${tag}_$leg.log 2>&1 || { echo "$leg failed"; tail -5 $root/gpurun_out/prof_${tag}_$leg.log; exit 1; }
  f=$(find $root/gpurun_out/prof_${tag}_$leg -name 'run_kernel_trace.csv' | head -1)
  { python3 $root/tools/trace_summary.py $f; grep "ms per step" $root/gpurun_out/prof_${tag}_$leg.log;
    echo "(rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/leg_step.py $leg 5; tools/trace_summary.py on the kernel trace: the last graph replay)"; } > $root/gpurun_out/${tag}_${leg}_step_summary.txt
  cat $root/gpurun_out/${tag}_${leg}_step_summary.txt
done
