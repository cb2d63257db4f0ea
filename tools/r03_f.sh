set -o pipefail
mkdir -p gpurun_out/r03f
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "aged_model or subsense_golden or qvga" > gpurun_out/r03f/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 gpurun_out/r03f/pytest.log
