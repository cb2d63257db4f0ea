#!/bin/bash
# ss_phase_a_kernel: dense I passes (BGS_SS_IPASS_MIN=n: a pass that fewer than n lanes would join waits for the next trip) against
# every pass at once (1, the round-3 form): step and kernel times per setting, then the instruction counters of two settings
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== BGS_SS_IPASS_MIN=$v"
  BGS_SS_IPASS_MIN=$v python3 $R/tools/bench_configs.py --only subsense8both 2>&1 | grep -h "SuBSENSE"
done
