set -o pipefail
mkdir -p gpurun_out/r02q
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "lbsp or subsense or lobster or sample_consensus or large_batches" > gpurun_out/r02q/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r02q/pytest.log
[ $rc -eq 0 ] || exit 1
for k in lbsp subsense8 subsense lobster; do timeout -k 10 300 python tools/bench_configs.py --only $k 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02q/bench.txt; done
