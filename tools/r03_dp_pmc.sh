#!/bin/bash
# round-3 scratch: HBM traffic of the dp GMM kernel per input
set -e
for leg in dpzsat dpzsurv dpgsat; do
  python tools/bench_configs.py --only $leg 2>&1 | grep DP
  bash tools/pmc_kernel.sh ${leg}_f dp_gmm FETCH_SIZE -- $GRAFT_REPO_ROOT/tools/bench_configs.py --only $leg
  bash tools/pmc_kernel.sh ${leg}_w dp_gmm WRITE_SIZE -- $GRAFT_REPO_ROOT/tools/bench_configs.py --only $leg
done
