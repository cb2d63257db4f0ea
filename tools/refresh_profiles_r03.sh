# Refresh round 3's evidence on one box (gpurun -- bash tools/refresh_profiles_r03.sh).  Outputs under gpurun_out/refresh3/, copied
# into profiles/ by hand: every bench_configs leg, the driver's bench command outside and inside rocprofv3 (+ counter passes),
# counters of the byte-stream kernels and of the fused group.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/refresh3; mkdir -p $O
( timeout -k 10 900 python tools/bench_configs.py; for k in byte mog1 subsense8 subsense8aged lobster pipeline dp cc clip clip1 clipdp clipfd group; do timeout -k 10 300 python tools/bench_configs.py --only $k; done ) 2>&1 | grep -v amdgpu.ids > $O/bench_configs.txt; echo "bench_configs rc=$?"
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_driver_args.json 2> $O/bench.err; echo "bench rc=$?"
bash tools/prof_r03.sh r03 20 5 > $O/prof_r03.log 2>&1; echo "prof rc=$?"
cp gpurun_out/prof_r03/*.json gpurun_out/prof_r03/kernel_stats.csv gpurun_out/prof_r03/launch_series.csv $O/ 2>/dev/null
{
  echo "# byte-stream kernels, 8 x 3840x2160 (tools/bench_configs.py --only byte): one rocprofv3 --pmc pass per counter group, mean per dispatch"
  bash tools/pmc_all.sh byte_f FETCH_SIZE -- $R/tools/bench_configs.py --only byte
  bash tools/pmc_all.sh byte_w WRITE_SIZE -- $R/tools/bench_configs.py --only byte
  bash tools/pmc_all.sh byte_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES" -- $R/tools/bench_configs.py --only byte
  echo "# WMV + ABL as one fused launch (--only group)"
  bash tools/pmc_all.sh grp_f FETCH_SIZE -- $R/tools/bench_configs.py --only group
  bash tools/pmc_all.sh grp_w WRITE_SIZE -- $R/tools/bench_configs.py --only group
} > $O/byte_kernels_pmc.txt 2>&1; echo "pmc rc=$?"
timeout -k 10 400 python tools/fuzz_parity.py 180 31000 > $O/fuzz_small.log 2>&1; echo "fuzz small rc=$?"
timeout -k 10 400 python tools/fuzz_parity.py 120 77000 big > $O/fuzz_big.log 2>&1; echo "fuzz big rc=$?"
{
  echo "# tools/fuzz_parity.py: random differential cases against the oracle (class, geometry, streams, frames, entry point, parameters, stream resets all drawn at random)"
  for f in small big; do
    echo "## $f: $(tail -1 $O/fuzz_$f.log)"
    grep '^\[' $O/fuzz_$f.log | awk '{print $2}' | sort | uniq -c | sort -rn
  done
} > $O/fuzz_parity.txt
tail -3 $O/bench_configs.txt; head -4 $O/fuzz_parity.txt
