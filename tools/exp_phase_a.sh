set -o pipefail
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "GMG or gmg" 2>&1 | tail -2
timeout -k 10 300 python tools/bench_configs.py 2>&1 | grep "GMG"
