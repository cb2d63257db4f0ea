#!/bin/bash
# ss_phase_a_kernel, dense I passes (BGS_SS_IPASS_MIN, default 16) against every pass at once (1): instruction and wait counters, young and aged model
R=$GRAFT_REPO_ROOT
for v in 1 16; do
  for leg in subsense8 subsense8aged1; do
    echo "== BGS_SS_IPASS_MIN=$v  $leg"
    BGS_SS_IPASS_MIN=$v bash $R/tools/pmc_kernel.sh ip$v ss_phase_a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" -- $R/tools/bench_configs.py --only $leg
    grep -h SuBSENSE $R/gpurun_out/pmc_ip$v/run.log
  done
done
