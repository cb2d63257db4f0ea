#!/bin/bash
# The driver line's SuBSENSE block (moving objects: 12 % / 7 % foreground) for a few settings of BGS_SS_IPASS_MIN, alternating on one box
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== BGS_SS_IPASS_MIN=$v"
  BGS_SS_IPASS_MIN=$v python3 $R/tools/bench_configs.py --only driverconfigs 2>&1 | grep -h "driver configs3"
done
