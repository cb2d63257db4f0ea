set -o pipefail
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "morphology or floodfill or subsense_golden or subsense_qvga or subsense_large or lobster_golden or gmg or GMG" 2>&1 | tail -2
for k in subsense8 subsense; do timeout -k 10 300 python tools/bench_configs.py --only $k 2>&1 | grep -v amdgpu.ids; done
