set -o pipefail
mkdir -p gpurun_out/r02l
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "subsense or lobster or sample_consensus or large_batches or frozen" > gpurun_out/r02l/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r02l/pytest.log
[ $rc -eq 0 ] || exit 1
for park in 1 24; do echo "== park $park"; BGS_SS_PARK=$park timeout -k 10 300 python tools/bench_configs.py --only subsense8 2>&1 | grep SuBSENSE | tee -a gpurun_out/r02l/bench.txt; done
for k in subsense lobster pipeline; do timeout -k 10 300 python tools/bench_configs.py --only $k 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02l/bench.txt; done
