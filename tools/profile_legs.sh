#!/bin/bash
# rocprofv3 kernel traces of the detector (cfg4) and PRN (cfg5) train steps -> gpurun_out/prof_<tag>_{detector,prn}/ and the
# step summaries under profiles/: tools/profile_legs.sh <tag>       (run on the GPU box; the program sits directly behind `--`)
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for leg in detector prn; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_${tag}_$leg -o run -- python3 $root/tools/leg_step.py $leg 5 > $root/gpurun_out/prof_${tag}_$leg.log 2>&1 || { echo "$leg failed"; tail -5 $root/gpurun_out/prof_${tag}_$leg.log; exit 1; }
  f=$(find $root/gpurun_out/prof_${tag}_$leg -name 'run_kernel_trace.csv' | head -1)
  { python3 $root/tools/trace_summary.py $f; tail -1 $root/gpurun_out/prof_${tag}_$leg.log;
    echo "(rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/leg_step.py $leg 5; tools/trace_summary.py on the kernel trace: the last graph replay)"; } > $root/profiles/${tag}_${leg}_step_summary.txt
  cp $(dirname $f)/run_kernel_stats.csv $root/profiles/${tag}_${leg}_kernel_stats.csv 2>/dev/null
  cat $root/profiles/${tag}_${leg}_step_summary.txt
done
