#!/bin/bash
# round-3 scratch: flood fill with log-step column fill - parity subset, then the aged SuBSENSE step (plain and per-kernel)
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r03o
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_00_configs.py -x -q -k "flood or subsense or lobster or morph or components or blobs" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python tools/bench_configs.py --only subsense8aged 2>&1 | grep SuBSENSE
python tools/bench_configs.py --only subsense8 2>&1 | grep SuBSENSE
cd /tmp && export TMPDIR=/tmp
export BGS_SS_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ss_aged -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only subsense8aged > $O/ss_aged.log 2>&1
