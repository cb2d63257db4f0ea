set -o pipefail
mkdir -p gpurun_out/exp_a
for lib in "" NOLOOP NOLOAD NOEXP NOLOADNOEXP; do
  echo "== ${lib:-product}" | tee -a gpurun_out/exp_a/out2.txt
  BGS_LIB_PATH=${lib:+$PWD/tracking_amd/lib/exp/lib_$lib.so} timeout -k 10 300 python tools/bench_configs.py --only subsense8 2>&1 | grep SuBSENSE | tee -a gpurun_out/exp_a/out2.txt
done
