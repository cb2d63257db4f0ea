#!/bin/bash
# ss_phase_b_kernel on the aged model (8 x 1080p): instruction, LDS, wait and memory counters
R=$GRAFT_REPO_ROOT
bash $R/tools/pmc_kernel.sh pb ss_phase_b "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" -- $R/tools/bench_configs.py --only subsense8aged1
bash $R/tools/pmc_kernel.sh pb ss_phase_b "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM" -- $R/tools/bench_configs.py --only subsense8aged1
bash $R/tools/pmc_kernel.sh pb ss_phase_b "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" -- $R/tools/bench_configs.py --only subsense8aged1
