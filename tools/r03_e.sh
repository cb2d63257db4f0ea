# SQ counters of the MOG2 filter kernel (who is issue-bound, who waits)
R=$GRAFT_REPO_ROOT
export BGS_MOG2_SPARSE=4
bash $R/tools/pmc_kernel.sh mog2_s4 "mog2_update_kernel<4>" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" -- $R/bench.py --gpus 1 --steps 20 --warmup 5 --main-only --no-pmc --no-cpu-baseline --sustain 0 --settle 100
bash $R/tools/pmc_kernel.sh mog2_s4b "mog2_update_kernel<4>" "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum" -- $R/bench.py --gpus 1 --steps 20 --warmup 5 --main-only --no-pmc --no-cpu-baseline --sustain 0 --settle 100
