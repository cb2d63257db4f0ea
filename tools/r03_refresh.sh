#!/bin/bash
# round-3 scratch: staged refresh kernel - parity subset, then the initial fill's duration from a kernel trace
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r03s
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_00_configs.py tests/test_gpu_05_lifecycle.py -x -q -k "subsense or SuBSENSE" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ss8 -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only subsense8 > $O/ss8.log 2>&1
grep refresh $O/ss8/t_kernel_stats.csv
