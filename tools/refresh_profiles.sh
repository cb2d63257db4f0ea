# Refresh the round's evidence on one box (gpurun -- bash tools/refresh_profiles.sh): bench_configs lines, SuBSENSE per-kernel stats,
# the driver's bench command inside and outside rocprofv3.  Outputs under gpurun_out/refresh/, copied into profiles/ by hand.
set -o pipefail
O=gpurun_out/refresh; mkdir -p $O
( timeout -k 10 900 python tools/bench_configs.py; for k in subsense8 subsense8aged lobster pipeline dp cc clip clip1 clipdp clipfd byte32; do timeout -k 10 300 python tools/bench_configs.py --only $k; done ) 2>&1 | grep -v amdgpu.ids > $O/bench_configs.txt; echo "bench_configs rc=$?"
bash tools/prof_subsense8.sh ss8 > $O/prof_ss8.log 2>&1; cp gpurun_out/prof_ss8/kernel_stats.csv $O/subsense_kernel_stats.csv; echo "ss8 rc=$?"
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_driver_args.json 2> $O/bench.err; echo "bench rc=$?"
bash tools/prof_r02.sh r02 20 5 > $O/prof_r02.log 2>&1; echo "prof rc=$?"
cp gpurun_out/prof_r02/*.json gpurun_out/prof_r02/kernel_stats.csv gpurun_out/prof_r02/launch_series.csv $O/ 2>/dev/null
tail -3 $O/bench_configs.txt
