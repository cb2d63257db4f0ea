#!/bin/bash
# round-3 scratch: WMM / WMV integer fast paths - parity (incl. all 2^24 triples), then the byte-stream timings
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r03q
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_06_group.py tests/test_gpu_00_configs.py tests/test_gpu_03_clip.py tests/test_gpu_01_host_cpp.py -x -q -k "wmv or wmm or Weighted or golden or seeded or group or 4k or device_batch or clip or demo or host" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python tools/bench_configs.py --only byte 2>&1 | grep -i "Weighted"
