#!/bin/bash
# Round 2: profile the DRIVER's exact bench command - kernel-trace stats in one run, FETCH_SIZE / WRITE_SIZE
# in their own --pmc runs - and summarise the timed region (K launches before the sustained leg) and the sustained leg.
# Usage: tools/prof_r02.sh <tag> <K> <W>
set -e
TAG=${1:-r02}; K=${2:-20}; W=${3:-5}; SUS=200
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--gpus 1 --steps $K --warmup $W --main-only --sustain $SUS"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $R/bench.py $ARGS --series $OUT/launch_series.csv > $OUT/stats.log 2>&1
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $R/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $R/bench.py $ARGS > $OUT/pmc_write.log 2>&1
echo "write pass done"
python3 $R/tools/prof_summary.py $OUT mog2_update 10000000 13669171200 $K $SUS > $OUT/mog2_timed_summary.json 2>&1 || true
python3 $R/tools/prof_summary.py $OUT mog2_update 10000000 13669171200 $SUS 0 > $OUT/mog2_sustained_summary.json 2>&1 || true
cp $OUT/stats/stats_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || true
grep -h '^{' $OUT/stats.log > $OUT/bench_line_under_rocprof.json 2>/dev/null || true
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
cat $OUT/mog2_timed_summary.json | head -12
