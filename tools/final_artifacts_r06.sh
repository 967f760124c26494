#!/bin/bash
# Round-6 final artifacts, two steps:
#   on the GPU box (gpurun):  tools/final_artifacts_r06.sh run      -> gpurun_out/bench_r06_final.json, prof_r06f/, r06f leg traces
#   afterwards, locally:      tools/final_artifacts_r06.sh collect  -> profiles/r06_*
if [ "$1" = "run" ]; then
  root=${GRAFT_REPO_ROOT:-/root/repo}
  cd $root && python bench.py > gpurun_out/bench_r06_final.json 2> gpurun_out/bench_r06_final.err || exit 1
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_r06f -o run -- python3 $root/bench.py --no-cpu-baseline > $root/gpurun_out/prof_r06f.log 2>&1 || exit 1
  cd $root && tools/profile_legs.sh r06f > gpurun_out/r06f_legs.txt 2>&1
  # the dominant kernel's counters: one rocprofv3 --pmc pass per counter (MI355X_MICROARCH.md, HBM / rocprofv3 PMC slots)
  cd $root && tools/pmc_passes.sh r06t 'FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE' tools/one_conv.py fwd 128 128 128 3 > gpurun_out/r06t_pmc.txt 2>&1
  exit 0
fi
f=$(find gpurun_out/prof_r06f -name run_kernel_trace.csv | head -1)
python tools/trace_summary.py $f > /tmp/sum.txt
python tools/dominant_from_trace.py $f >> /tmp/sum.txt
python - <<'PY' >> /tmp/sum.txt
import json, re
d = json.loads(open('gpurun_out/bench_r06_final.json').read().strip().splitlines()[-1])
u = json.loads(re.search(r'\{"metric".*\}', open('gpurun_out/prof_r06f.log').read()).group(0))
print(f"(rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline; the bench line printed under "
      f"the profiler in that run: ms_per_step {u['ms_per_step']}, roofline.launch_us {u['roofline']['launch_us']}; the unprofiled default "
      f"run before it on the same box: ms_per_step {d['ms_per_step']}, value {d['value']}, roofline.frac {d['roofline']['frac']}, "
      f"launch_us {d['roofline']['launch_us']}; tools/trace_summary.py + tools/dominant_from_trace.py on the kernel trace)")
PY
cp /tmp/sum.txt profiles/r06_final_step_summary.txt
cp $(dirname $f)/run_kernel_stats.csv profiles/r06_final_kernel_stats.csv
cp gpurun_out/bench_r06_final.json profiles/r06_bench_default.json
tools/profile_legs.sh --collect r06f
python tools/make_traffic_json.py r06 r06t
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  f2=$(find gpurun_out/pmc_r06t/$c -name '*counter_collection.csv' | head -1)
  [ -n "$f2" ] && grep -E "Counter_Name|conv3x3" $f2 > profiles/r06_conv3x3_pmc_$c.csv
done
cat profiles/r06_final_step_summary.txt
