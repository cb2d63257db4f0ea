#!/bin/bash
# round-3 scratch: A/B of two builds of the library on the aged SuBSENSE step (phase B on the main stream so that wall = sum of kernels)
set -e
for ov in 0 1; do
  echo "== BGS_SS_OVERLAP=$ov, in-tree build"
  BGS_SS_OVERLAP=$ov python tools/bench_configs.py --only subsense8aged 2>&1 | grep SuBSENSE
  echo "== BGS_SS_OVERLAP=$ov, A/B build"
  BGS_SS_OVERLAP=$ov BGS_LIB_PATH=$PWD/tracking_amd/lib/ab/libbgs_hip.so python tools/bench_configs.py --only subsense8aged 2>&1 | grep SuBSENSE
done
