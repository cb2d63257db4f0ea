#!/bin/bash
# Per-kernel breakdown of the 8-stream SuBSENSE step (S_surv): rocprofv3 kernel-trace stats, bgs:: kernels only.
set -e
TAG=${1:-ss8}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only subsense8 > $OUT/stats.log 2>&1
grep -E '^"Name"|bgs::' $OUT/stats/stats_kernel_stats.csv | cut -c1-160 > $OUT/kernel_stats.csv
rm -rf $OUT/stats
grep -v amdgpu $OUT/stats.log | tail -2
cat $OUT/kernel_stats.csv
