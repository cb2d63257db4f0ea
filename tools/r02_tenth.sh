set -o pipefail
mkdir -p gpurun_out/r02k
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "subsense or lobster or sample_consensus or large_batches or frozen" > gpurun_out/r02k/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r02k/pytest.log
[ $rc -eq 0 ] || exit 1
for lib in "" AHEAD1 AHEAD2; do for park in 1 24; do echo "== ahead ${lib:-3(product)} park $park"; BGS_SS_PARK=$park BGS_LIB_PATH=${lib:+$PWD/tracking_amd/lib/exp/lib_$lib.so} timeout -k 10 300 python tools/bench_configs.py --only subsense8 2>&1 | grep SuBSENSE | tee -a gpurun_out/r02k/bench.txt; done; done
for k in subsense lobster pipeline; do timeout -k 10 300 python tools/bench_configs.py --only $k 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02k/bench.txt; done
