set -o pipefail
mkdir -p gpurun_out/r02d
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02d/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r02d/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/bench_configs.py --only byte 2>&1 | grep Adaptive | tee gpurun_out/r02d/byte.txt
timeout -k 10 300 python tools/bench_configs.py --only subsense 2>&1 | tee gpurun_out/r02d/subsense.txt
timeout -k 10 300 python tools/bench_configs.py --only subsense8 2>&1 | tee -a gpurun_out/r02d/subsense.txt
timeout -k 10 300 python tools/bench_configs.py --only pipeline 2>&1 | tee -a gpurun_out/r02d/subsense.txt
