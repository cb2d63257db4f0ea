set -o pipefail
mkdir -p gpurun_out/r02i
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "subsense or sample_consensus or large_batches" > gpurun_out/r02i/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r02i/pytest.log
[ $rc -eq 0 ] || exit 1
for park in 8 16 24 32 48 64; do echo "park $park"; BGS_SS_PARK=$park timeout -k 10 300 python tools/bench_configs.py --only subsense8 2>&1 | grep SuBSENSE | tee -a gpurun_out/r02i/bench.txt; done
for refill in 8 32; do echo "refill $refill park 32"; BGS_SS_REFILL=$refill BGS_SS_PARK=32 timeout -k 10 300 python tools/bench_configs.py --only subsense8 2>&1 | grep SuBSENSE | tee -a gpurun_out/r02i/bench.txt; done
timeout -k 10 300 python tools/bench_configs.py --only subsense 2>&1 | grep SuBSENSE | tee -a gpurun_out/r02i/bench.txt
