#!/bin/bash
# Per-kernel breakdown of a SuBSENSE run (2 x 1080p, S_surv then smooth): rocprofv3 kernel-trace stats, summarised on the box.
set -e
TAG=${1:-ss}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only subsense > $OUT/stats.log 2>&1
cp $OUT/stats/stats_kernel_stats.csv $OUT/kernel_stats.csv
rm -rf $OUT/stats
grep -v amdgpu $OUT/stats.log | tail -3
head -30 $OUT/kernel_stats.csv | cut -c1-200
