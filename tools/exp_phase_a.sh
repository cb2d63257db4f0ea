set -o pipefail
for k in clip8sat clip8surv; do
bash tools/pmc_kernel.sh ${k}_a mog2_clip "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" -- $GRAFT_REPO_ROOT/tools/bench_configs.py --only $k
bash tools/pmc_kernel.sh ${k}_b mog2_clip "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" -- $GRAFT_REPO_ROOT/tools/bench_configs.py --only $k
done
