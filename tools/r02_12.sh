set -o pipefail
mkdir -p gpurun_out/r02m
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "subsense or lobster or sample_consensus or large_batches or frozen" > gpurun_out/r02m/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r02m/pytest.log
[ $rc -eq 0 ] || exit 1
for k in subsense8 subsense pipeline; do timeout -k 10 300 python tools/bench_configs.py --only $k 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02m/bench.txt; done
bash tools/pmc_kernel.sh pc ss_phase_a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS" -- $GRAFT_REPO_ROOT/tools/bench_configs.py --only subsense8 2>&1 | tee gpurun_out/r02m/pmc.txt
