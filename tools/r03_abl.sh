#!/bin/bash
# round-3 scratch: abl_kernel with software prefetch - parity subset, timings
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r03t
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_06_group.py tests/test_gpu_00_configs.py tests/test_gpu_05_lifecycle.py tests/test_gpu_03_clip.py -x -q -k "abl or Adaptive or golden or seeded or group or 4k or device_batch or lifecycle or clip" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python tools/bench_configs.py --only byte 2>&1 | grep -i "AdaptiveBackground"
python tools/bench_configs.py --only group 2>&1 | grep -i "group"
