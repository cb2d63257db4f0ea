# Refresh round 4's evidence on one box (gpurun -- bash tools/refresh_profiles_r04.sh).  Outputs under gpurun_out/refresh4/, copied into
# profiles/ by hand: the driver's bench command outside and inside rocprofv3 (+ counter passes), every bench_configs leg, the A/B of the
# MOG2 kernel against round 3's, the random differential run.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/refresh4; mkdir -p $O
timeout -k 10 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_driver_args.json 2> $O/bench.err; echo "bench rc=$?"
bash tools/prof_r04.sh r04 20 5 > $O/prof_r04.log 2>&1; echo "prof rc=$?"
cp gpurun_out/prof_r04/*.json gpurun_out/prof_r04/kernel_stats.csv gpurun_out/prof_r04/launch_series.csv $O/ 2>/dev/null
bash tools/r04_abmog2.sh > $O/abmog2.txt 2>&1; echo "ab rc=$?"
( timeout -k 10 900 python tools/bench_configs.py; for k in byte mog1 subsense8 subsense8aged lobster pipeline dp cc clip clip1 clipdp clipfd group gmg; do timeout -k 10 300 python tools/bench_configs.py --only $k; done ) 2>&1 | grep -v amdgpu.ids > $O/bench_configs.txt; echo "bench_configs rc=$?"
timeout -k 10 400 python tools/fuzz_parity.py 180 41000 > $O/fuzz_small.log 2>&1; echo "fuzz small rc=$?"
timeout -k 10 400 python tools/fuzz_parity.py 120 87000 big > $O/fuzz_big.log 2>&1; echo "fuzz big rc=$?"
{
  echo "# tools/fuzz_parity.py: random differential cases against the oracle (class, geometry, streams, frames, entry point, parameters, stream resets all drawn at random)"
  for f in small big; do
    echo "## $f: $(tail -1 $O/fuzz_$f.log)"
    grep '^\[' $O/fuzz_$f.log | awk '{print $2}' | sort | uniq -c | sort -rn
  done
} > $O/fuzz_parity.txt
tail -3 $O/bench_configs.txt; head -4 $O/fuzz_parity.txt
