#!/bin/bash
# PMC of ss_phase_a_kernel on 8 x 1080p S_surv (current build)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_pa
: > $R/gpurun_out/pmc_pa/summary3.txt
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "FETCH_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH"; do
  echo "== $ctrs" | tee -a $R/gpurun_out/pmc_pa/summary3.txt
  bash $R/tools/pmc_kernel.sh pd ss_phase_a "$ctrs" -- $R/tools/bench_configs.py --only subsense8 2>&1 | tee -a $R/gpurun_out/pmc_pa/summary3.txt
  grep -h SuBSENSE $R/gpurun_out/pmc_pd/run.log | tee -a $R/gpurun_out/pmc_pa/summary3.txt
done
