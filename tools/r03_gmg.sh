#!/bin/bash
# round-3 scratch: GMG stores only changed colours - parity subset, timing, write traffic
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r03u
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "gmg or GMG" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python tools/bench_configs.py --only gmg 2>&1 | grep GMG
bash tools/pmc_kernel.sh gmg_w gmg_kernel WRITE_SIZE -- $GRAFT_REPO_ROOT/tools/bench_configs.py --only gmg
