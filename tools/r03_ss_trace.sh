#!/bin/bash
# round-3 scratch: per-kernel times of the aged SuBSENSE step with phase B on the main stream (exclusive kernel times)
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r03n
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export BGS_SS_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ss_aged -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only subsense8aged > $O/ss_aged.log 2>&1
grep SuBSENSE $O/ss_aged.log
