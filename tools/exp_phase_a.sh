set -o pipefail
mkdir -p gpurun_out/exp_a
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "subsense_golden or subsense_qvga or subsense_gray or subsense_ragged" 2>&1 | tail -2
for park in 1 4 8 16 24 32; do
  echo "== park $park" | tee -a gpurun_out/exp_a/out4.txt
  BGS_SS_PARK=$park timeout -k 10 300 python tools/bench_configs.py --only subsense8 2>&1 | grep SuBSENSE | tee -a gpurun_out/exp_a/out4.txt
done
