#!/bin/bash
# round-3 scratch: WMV integer fast path - parity (incl. all 2^24 triples), then the byte-stream timings at 4 and 16 pixels per lane
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r03q
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_06_group.py tests/test_gpu_00_configs.py tests/test_gpu_03_clip.py -x -q -k "wmv or Weighted or golden or seeded or group or 4k or device_batch or clip" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python tools/bench_configs.py --only byte 2>&1 | grep -i "variance"
BGS_FRAME_GROUP=16 python tools/bench_configs.py --only byte 2>&1 | grep -i "variance"
