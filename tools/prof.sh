#!/bin/bash
# Profile bench.py on the GPU box: kernel-trace stats in one run, PMC counters in their own runs (gpurun refuses
# --pmc combined with trace domains other than kernel-trace).  Usage: tools/prof.sh <tag> [bench args...]
set -e
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 200 --warmup 50 --no-cpu-baseline --main-only $@"   # bench.py defaults; the last 200 big launches are the timed region
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -o tcc -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_tcc.log 2>&1 || true
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o sq -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_sq.log 2>&1 || true
# summarise on the box and keep only what is small (gpurun merges at most 64 MiB back)
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT mog2_update 10000000 13669171200 200 > $OUT/mog2_summary.json 2>&1 || true
cp $OUT/stats/stats_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || true
grep -h "placement probe" $OUT/*.log > $OUT/probe.txt 2>/dev/null || true
grep -h '^{' $OUT/stats.log > $OUT/bench_line_under_rocprof.json 2>/dev/null || true
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_tcc $OUT/pmc_sq
ls -la $OUT
