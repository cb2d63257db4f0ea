#!/bin/bash
# What the waves of ss_phase_a_kernel's stage 2 do (a -DBGS_SS_STATS build of the library, tracking_amd/lib/stats: trips, R executions and
# I passes with the lanes that took part), young and aged model, for a few settings of BGS_SS_IPASS_MIN
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  for leg in subsense8 subsense8aged1; do
    echo "== BGS_SS_IPASS_MIN=$v $leg"
    BGS_SS_IPASS_MIN=$v BGS_LIB_PATH=$R/tracking_amd/lib/stats/libbgs_hip.so BGS_LIB_PARTIAL_ABI=1 python3 $R/tools/bench_configs.py --only $leg 2>&1 | grep -h "SuBSENSE\|ss_stats"
  done
done
