set -o pipefail
mkdir -p gpurun_out/r02t
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "AdaptiveSelective or asbl" > gpurun_out/r02t/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r02t/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/bench_configs.py --only byte 2>&1 | grep "Selective" | tee -a gpurun_out/r02t/bench.txt
