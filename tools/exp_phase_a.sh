set -o pipefail
bash tools/pmc_kernel.sh lob_s lob_phase_a "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" -- $GRAFT_REPO_ROOT/tools/bench_configs.py --only lobster
bash tools/pmc_kernel.sh lob_q lob_phase_a "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" -- $GRAFT_REPO_ROOT/tools/bench_configs.py --only lobster
bash tools/pmc_kernel.sh lob_f lob_phase_a "FETCH_SIZE" -- $GRAFT_REPO_ROOT/tools/bench_configs.py --only lobster
bash tools/pmc_kernel.sh lob_t lob_phase_a "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" -- $GRAFT_REPO_ROOT/tools/bench_configs.py --only lobster
