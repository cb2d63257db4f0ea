set -o pipefail
export BGS_LIB_PARTIAL_ABI=1
for lib in OLDPLANARFIX; do
  echo "== ${lib:-product}"
  export BGS_LIB_PATH=${lib:+$PWD/tracking_amd/lib/exp/lib_$lib.so}
  timeout -k 10 300 python tools/bench_configs.py --only subsense8 2>&1 | grep SuBSENSE
  bash tools/pmc_kernel.sh pe ss_phase_a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAVE_CYCLES" -- $GRAFT_REPO_ROOT/tools/bench_configs.py --only subsense8 2>&1
done
