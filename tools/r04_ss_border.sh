#!/bin/bash
# SuBSENSE step with phase B started right behind phase A (default) against behind the flood fill (BGS_SS_B_LATE=1):
# step wall time young / aged, alternating, then the timeline of one aged step each way
R=$GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  echo "== BGS_SS_B_LATE=$v"
  BGS_SS_B_LATE=$v python3 $R/tools/bench_configs.py --only subsense8both 2>&1 | grep -h "SuBSENSE" | sed 's/.*streams: //'
done
for v in 0 1; do
  echo "== timeline, BGS_SS_B_LATE=$v"
  BGS_SS_B_LATE=$v LEG=subsense8aged1 bash $R/tools/trace_ss_step.sh
done
