#!/bin/bash
# round-3 scratch: ASBL table kernel A/B
set -e
O=gpurun_out/r03m
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_06_group.py tests/test_gpu_00_configs.py tests/test_gpu_05_lifecycle.py -x -q -k "asbl or Adaptive or golden or seeded or group or 4k or device_batch or lifecycle" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python tools/bench_configs.py --only byte > $O/byte_table.txt 2>&1
grep -h Adaptive $O/byte_table.txt; grep -h ASBL $O/byte_table.txt
