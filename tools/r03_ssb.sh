#!/bin/bash
# round-3 scratch: phase B rewrite - parity subset, then the SuBSENSE / LOBSTER steps
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r03p
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_00_configs.py tests/test_gpu_05_lifecycle.py -x -q -k "subsense or lobster or SuBSENSE or LOBSTER" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python tools/bench_configs.py --only subsense8aged 2>&1 | grep SuBSENSE
python tools/bench_configs.py --only subsense8 2>&1 | grep SuBSENSE
python tools/bench_configs.py --only lobster 2>&1 | grep LOBSTER
cd /tmp && export TMPDIR=/tmp
export BGS_SS_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ss_aged -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only subsense8aged1 > $O/ss_aged.log 2>&1
