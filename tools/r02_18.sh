set -o pipefail
mkdir -p gpurun_out/r02s
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "WeightedMovingMean or unweighted or wrapper_threshold or single_channel or warmup" > gpurun_out/r02s/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r02s/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/bench_configs.py --only byte 2>&1 | grep "MovingMean" | tee -a gpurun_out/r02s/bench.txt
BGS_FRAME_GROUP=16 timeout -k 10 300 python tools/bench_configs.py --only byte 2>&1 | grep "MovingMean" | tee -a gpurun_out/r02s/bench.txt
