#!/bin/bash
# round-3 scratch: MOG1 slot layout - parity, timings, traffic counters
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r03r
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "mog1 or MOG1 or MixtureOfGaussianV1 or golden or seeded or single_channel or device_batch or clip or lifecycle or demo or type_table" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python tools/bench_configs.py --only mog1 2>&1 | grep Mixture
python tools/bench_configs.py --only clip1 2>&1 | grep MOG1
bash tools/pmc_kernel.sh m1_fetch mog1_update FETCH_SIZE -- $GRAFT_REPO_ROOT/tools/bench_configs.py --only mog1sat
bash tools/pmc_kernel.sh m1_write mog1_update WRITE_SIZE -- $GRAFT_REPO_ROOT/tools/bench_configs.py --only mog1sat
bash tools/pmc_kernel.sh m1s_fetch mog1_update FETCH_SIZE -- $GRAFT_REPO_ROOT/tools/bench_configs.py --only mog1surv
bash tools/pmc_kernel.sh m1s_write mog1_update WRITE_SIZE -- $GRAFT_REPO_ROOT/tools/bench_configs.py --only mog1surv
