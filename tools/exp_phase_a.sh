set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "subsense or lobster or sample_consensus or large_batches or disjoint or flood" 2>&1 | tail -3
bash tools/prof_subsense8.sh ss8
