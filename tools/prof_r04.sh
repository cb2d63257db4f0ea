#!/bin/bash
# Round 4 (as round 3): profile the DRIVER's exact bench command - kernel-trace stats in one run, FETCH_SIZE / WRITE_SIZE / SQ counters in their
# own --pmc runs - and summarise the timed region (K launches before the sustained leg) and the sustained leg of the MOG2 kernel.
# Usage: tools/prof_r03.sh <tag> <K> <W>
set -e
TAG=${1:-r04}; K=${2:-20}; W=${3:-5}; SUS=200
ALGO=5971968000   # 90 B/pixel x 32 x 1920 x 1080 (bench.py BYTES_PER_PIXEL)
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
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o sq -- python3 $R/bench.py $ARGS > $OUT/pmc_sq.log 2>&1
echo "sq pass done"
python3 $R/tools/prof_summary.py $OUT mog2_update 10000000 $ALGO $K $SUS > $OUT/mog2_timed_summary.json 2>&1 || true
python3 $R/tools/prof_summary.py $OUT mog2_update 10000000 $ALGO $SUS 0 > $OUT/mog2_sustained_summary.json 2>&1 || true
cp $OUT/stats/stats_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || true
grep -h '^{' $OUT/stats.log > $OUT/bench_line_under_rocprof.json 2>/dev/null || true
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
cat $OUT/mog2_timed_summary.json | head -40
