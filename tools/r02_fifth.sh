set -o pipefail
mkdir -p gpurun_out/r02e
timeout -k 10 900 python -m pytest tests/test_gpu_01_host_cpp.py tests/test_gpu_parity.py -m gpu -x -q -k "blob or components or ustc or demo" > gpurun_out/r02e/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -30 gpurun_out/r02e/pytest.log
