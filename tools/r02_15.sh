set -o pipefail
mkdir -p gpurun_out/r02p
{ timeout -k 10 900 python tools/bench_configs.py; for k in dp cc pipeline lobster subsense8 byte; do timeout -k 10 300 python tools/bench_configs.py --only $k; done; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02p/bench_configs.txt
echo "bench_configs done"
bash tools/prof_any.sh r02allk tools/bench_configs.py > gpurun_out/r02p/prof_any.log 2>&1; echo "prof_any rc=$?"
ls gpurun_out/prof_r02allk | head
