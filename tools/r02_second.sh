set -o pipefail
mkdir -p gpurun_out/r02b
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 --series gpurun_out/r02b/series_k20.csv > gpurun_out/r02b/bench_k20.json 2> gpurun_out/r02b/bench_k20.err || { tail -20 gpurun_out/r02b/bench_k20.err; exit 1; }
cat gpurun_out/r02b/bench_k20.json
timeout -k 10 300 python bench.py --main-only --no-cpu-baseline --series gpurun_out/r02b/series_k200.csv > gpurun_out/r02b/bench_k200.json 2> gpurun_out/r02b/bench_k200.err || exit 1
timeout -k 10 300 python bench.py --main-only --steps 20 --warmup 5 --settle 0 --series gpurun_out/r02b/series_nosettle.csv > gpurun_out/r02b/bench_nosettle.json 2> gpurun_out/r02b/bench_nosettle.err || exit 1
timeout -k 10 600 bash tools/prof_r02.sh r02 20 5
