set -o pipefail
mkdir -p gpurun_out/r02c
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "abl or Adaptive or wrapper_threshold or device_batch or 4k" > gpurun_out/r02c/pytest_abl.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02c/pytest_abl.log
timeout -k 10 300 python tools/bench_configs.py --only byte 2>&1 | tee gpurun_out/r02c/byte_default.txt
BGS_FRAME_GROUP=4 timeout -k 10 300 python tools/bench_configs.py --only byte 2>&1 | grep Adaptive | tee gpurun_out/r02c/byte_g4.txt
