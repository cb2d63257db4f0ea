set -o pipefail
mkdir -p gpurun_out/r02f
timeout -k 10 900 python -m pytest tests/test_gpu_02_ingest.py -m gpu -x -q > gpurun_out/r02f/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -30 gpurun_out/r02f/pytest.log
