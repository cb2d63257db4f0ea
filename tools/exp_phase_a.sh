set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_03_clip.py -x -q 2>&1 | tail -4
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "dp or DP or mog1 or MixtureOfGaussianV1" 2>&1 | tail -3
timeout -k 10 600 python tools/bench_configs.py --only clipdp 2>&1 | grep -v amdgpu.ids
timeout -k 10 600 python tools/bench_configs.py --only dp 2>&1 | grep -v amdgpu.ids
