#!/bin/bash
# ss_phase_a_kernel with channel-split I passes (BGS_SS_SPLIT_MAX=42) against one candidate per lane (0): instruction and LDS counters
R=$GRAFT_REPO_ROOT
for v in 0 42; do
  echo "== BGS_SS_SPLIT_MAX=$v"
  BGS_SS_SPLIT_MAX=$v bash $R/tools/pmc_kernel.sh sp$v ss_phase_a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" -- $R/tools/bench_configs.py --only subsense8
  grep -h SuBSENSE $R/gpurun_out/pmc_sp$v/run.log
done
