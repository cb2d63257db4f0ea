#!/bin/bash
# PMC of ss_phase_a_kernel on 8 x 1080p S_surv, with and without parking of the inter-LBSP step
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_pa
: > $R/gpurun_out/pmc_pa/summary2.txt
for park in 1 24; do
  export BGS_SS_PARK=$park
  for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "FETCH_SIZE"; do
    echo "== park $park: $ctrs" | tee -a $R/gpurun_out/pmc_pa/summary2.txt
    bash $R/tools/pmc_kernel.sh pb ss_phase_a "$ctrs" -- $R/tools/bench_configs.py --only subsense8 2>&1 | tee -a $R/gpurun_out/pmc_pa/summary2.txt
    grep -h SuBSENSE $R/gpurun_out/pmc_pb/run.log | tee -a $R/gpurun_out/pmc_pa/summary2.txt
  done
done
