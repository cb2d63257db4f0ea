set -o pipefail
mkdir -p gpurun_out/r03k
bash tools/validate_gpu.sh
cp gpurun_out/validate/bench_k20.json gpurun_out/r03k/bench_k20.json
timeout -k 10 900 python tools/bench_configs.py > gpurun_out/r03k/bench_configs.txt 2>&1; echo "bench_configs rc=$?"
timeout -k 10 600 python tools/bench_configs.py --only clip >> gpurun_out/r03k/bench_configs.txt 2>&1; echo "clip rc=$?"
timeout -k 10 600 python tools/bench_configs.py --only subsense8 >> gpurun_out/r03k/bench_configs.txt 2>&1
timeout -k 10 600 python tools/bench_configs.py --only subsense8aged >> gpurun_out/r03k/bench_configs.txt 2>&1
timeout -k 10 600 python tools/bench_configs.py --only dp >> gpurun_out/r03k/bench_configs.txt 2>&1
timeout -k 10 600 python tools/bench_configs.py --only pipeline >> gpurun_out/r03k/bench_configs.txt 2>&1
tail -40 gpurun_out/r03k/bench_configs.txt | cut -c1-260
