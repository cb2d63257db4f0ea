set -o pipefail
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "floodfill or subsense or sample_consensus" 2>&1 | tail -2
timeout -k 10 300 python tools/exp_floodflags.py 2>&1 | grep -v amdgpu
for k in subsense8 subsense pipeline; do timeout -k 10 300 python tools/bench_configs.py --only $k 2>&1 | grep -v amdgpu.ids; done
