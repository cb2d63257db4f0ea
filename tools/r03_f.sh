set -o pipefail
mkdir -p gpurun_out/r03f
timeout -k 10 900 python -m pytest tests/test_gpu_01_host_cpp.py tests/test_gpu_06_group.py -m gpu -x -q > gpurun_out/r03f/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -30 gpurun_out/r03f/pytest.log
