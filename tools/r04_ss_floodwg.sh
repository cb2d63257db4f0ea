#!/bin/bash
# SuBSENSE step with the flood fill's strip kernel in 256-lane workgroups (default) against round 3's 1024-lane ones (BGS_SS_FLOOD_WG1024=1),
# which wait out phase B on the side stream: step wall time young / aged, then the timeline of one aged step each way
R=$GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  echo "== BGS_SS_FLOOD_WG1024=$v"
  BGS_SS_FLOOD_WG1024=$v python3 $R/tools/bench_configs.py --only subsense8both 2>&1 | grep -h "SuBSENSE" | sed 's/.*streams: //'
done
for v in 0 1; do
  echo "== timeline, BGS_SS_FLOOD_WG1024=$v"
  BGS_SS_FLOOD_WG1024=$v LEG=subsense8aged1 bash $R/tools/trace_ss_step.sh
done
