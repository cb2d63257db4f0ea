#!/bin/bash
# rocprofv3 kernel-trace stats of any python script: tools/prof_any.sh <tag> <script> [args...]; prints the top kernels
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 "$@" > $OUT/stats.log 2>&1
cp $OUT/stats/stats_kernel_stats.csv $OUT/kernel_stats.csv
rm -rf $OUT/stats
grep -v "amdgpu\|^W\|^E" $OUT/stats.log | tail -5
head -16 $OUT/kernel_stats.csv | cut -c1-170
