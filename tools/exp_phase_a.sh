set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "subsense or sample_consensus or large_batches or disjoint" 2>&1 | tail -3
for k in subsense8 subsense; do timeout -k 10 300 python tools/bench_configs.py --only $k 2>&1 | grep -v amdgpu.ids; done
