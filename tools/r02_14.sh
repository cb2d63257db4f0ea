set -o pipefail
mkdir -p gpurun_out/r02o
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "lobster or sample_consensus" > gpurun_out/r02o/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r02o/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/bench_configs.py --only lobster 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02o/bench.txt
