#!/bin/bash
# Per-kernel stats of the device pipeline leg (SuBSENSE -> connected components): rocprofv3 --kernel-trace --stats, bgs:: kernels only.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_pipe
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only pipeline > $OUT/stats.log 2>&1
grep -E '^"Name"|cc_|mask_' $OUT/stats/stats_kernel_stats.csv | cut -c1-150
rm -rf $OUT/stats
grep -a pipeline $OUT/stats.log
