#!/bin/bash
# SQ counters of the byte-stream kernels on 8 x 3840x2160 (tools/bench_configs.py --only byte): who is issue-bound, who waits
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_byte
: > $R/gpurun_out/pmc_byte/summary.txt
for k in asbl_kernel wmm_kernel wmv_kernel sigmadelta_kernel abl_kernel framediff_kernel; do
  echo "== $k" | tee -a $R/gpurun_out/pmc_byte/summary.txt
  bash $R/tools/pmc_kernel.sh byte_$k $k "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" -- $R/tools/bench_configs.py --only byte 2>&1 | tee -a $R/gpurun_out/pmc_byte/summary.txt
done
