set -o pipefail
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "SigmaDelta or sigma" 2>&1 | tail -3
timeout -k 10 600 python tools/bench_configs.py --only byte 2>&1 | grep -v amdgpu.ids | grep -i sigma
