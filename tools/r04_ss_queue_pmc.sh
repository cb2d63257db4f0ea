#!/bin/bash
# ss_phase_a_kernel: per-wave candidate queue (BGS_SS_QUEUE=1) against one candidate per lane and pass (0): instruction, LDS and wait counters
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  echo "== BGS_SS_QUEUE=$v"
  BGS_SS_QUEUE=$v bash $R/tools/pmc_kernel.sh sq$v ss_phase_a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" -- $R/tools/bench_configs.py --only subsense8
  BGS_SS_QUEUE=$v bash $R/tools/pmc_kernel.sh sq$v ss_phase_a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" -- $R/tools/bench_configs.py --only subsense8
  grep -h SuBSENSE $R/gpurun_out/pmc_sq$v/run.log
done
