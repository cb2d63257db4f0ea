# SQ counters of mog2_update_kernel: filter path vs eager path (who is issue-bound, who waits)
R=$GRAFT_REPO_ROOT
for m in 4 1; do
  export BGS_MOG2_SPARSE=$m
  echo "== BGS_MOG2_SPARSE=$m"
  bash $R/tools/pmc_kernel.sh mog2_s$m mog2_update "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" -- $R/bench.py --gpus 1 --steps 20 --warmup 5 --main-only --no-pmc --no-cpu-baseline --sustain 0 --settle 100
  bash $R/tools/pmc_kernel.sh mog2_s${m}b mog2_update "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum" -- $R/bench.py --gpus 1 --steps 20 --warmup 5 --main-only --no-pmc --no-cpu-baseline --sustain 0 --settle 100
done
