#!/bin/bash
# LOBSTER phase A: lanes fed from a queue (default) against one pixel per lane in lock step (BGS_LOB_QUEUE=0), 8 x 1080p, alternating on one box
R=$GRAFT_REPO_ROOT
for v in 1 0 1 0; do
  echo "== BGS_LOB_QUEUE=$v"
  BGS_LOB_QUEUE=$v python3 $R/tools/bench_configs.py --only lobster 2>&1 | grep -h "LOBSTER"
done
for rf in 4 8 32; do
  echo "== BGS_LOB_QUEUE=1 BGS_LOB_REFILL=$rf"
  BGS_LOB_REFILL=$rf python3 $R/tools/bench_configs.py --only lobster 2>&1 | grep -h "LOBSTER"
done
