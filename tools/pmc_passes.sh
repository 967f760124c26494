#!/bin/bash
# One rocprofv3 --pmc pass per counter (gpurun refuses --pmc next to the trace domains other than --kernel-trace):
#   tools/pmc_passes.sh <out tag> "<counter> <counter> ..." <script> <args ...>     (a,b,c = one pass with several counters)
# writes gpurun_out/pmc_<tag>/<counter>/..., then prints tools/pmc_summary.py of every pass.
tag=$1; counters=$2; shift 2
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out/pmc_$tag
for c in $counters; do
  rocprofv3 --pmc ${c//,/ } --kernel-trace --output-format csv -d $root/gpurun_out/pmc_$tag/$c -o run -- python3 $root/$1 "${@:2}" > $root/gpurun_out/pmc_$tag/$c.log 2>&1 || { echo "pass $c failed"; tail -5 $root/gpurun_out/pmc_$tag/$c.log; }
done
