#!/bin/bash
# Round 6, on the GPU box: a soak of the train step with the board power beside it, and the counter passes of the two north-star families
# (depthwise 128ch @128^2 forward; pointwise 512 -> 512 @32^2 forward) -> gpurun_out/r6/{soak.txt,soak_power.txt}, gpurun_out/pmc_r06dw, pmc_r06pw
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root && mkdir -p gpurun_out/r6
(python tools/soak.py 4000 > gpurun_out/r6/soak.txt 2>&1 &)
sleep 12
for i in $(seq 1 20); do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d "\n" >> gpurun_out/r6/soak_power.txt; echo >> gpurun_out/r6/soak_power.txt; sleep 1; done
wait
sleep 15
cat gpurun_out/r6/soak.txt
tools/pmc_passes.sh r06dw 'FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_INSTS_VALU,SQ_ACTIVE_INST_VALU,SQ_WAVE_CYCLES' tools/one_dw.py 128 128 1 > gpurun_out/r6/pmc_dw.txt 2>&1
tools/pmc_passes.sh r06pw 'FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES' tools/one_conv.py fwd 32 512 512 1 > gpurun_out/r6/pmc_pw.txt 2>&1
for d in r06dw r06pw; do echo "== $d"; for c in gpurun_out/pmc_$d/*/; do python tools/pmc_summary.py "" $c 2>/dev/null | head -12; done; done
