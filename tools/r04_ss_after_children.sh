#!/bin/bash
# Does the SuBSENSE block slow down after OTHER processes have used the GPU?  (bench.py with live PMC passes reports the aged step at
# 2.1 ms, without them at 1.67.)  The block in a fresh process; then after a MOG2 process has come and gone; then after one under rocprofv3 --pmc.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
echo "== fresh"; python3 $R/tools/bench_configs.py --only driverconfigs 2>&1 | grep "driver configs3"
python3 $R/bench.py --steps 5 --warmup 2 --main-only --no-pmc --no-cpu-baseline --settle 20 --sustain 5 > /dev/null 2>&1
echo "== after a plain MOG2 bench process"; python3 $R/tools/bench_configs.py --only driverconfigs 2>&1 | grep "driver configs3"
timeout -k 5 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcx -o pmc -- python3 $R/bench.py --steps 5 --warmup 2 --main-only --no-pmc --no-cpu-baseline --settle 20 --sustain 5 > /dev/null 2>&1
rm -rf /tmp/pmcx
echo "== after a MOG2 bench process under rocprofv3 --pmc"; python3 $R/tools/bench_configs.py --only driverconfigs 2>&1 | grep "driver configs3"
