#!/bin/bash
# round-3 scratch: the MOG2 headline with the library of the start of this session against the current one, same box, alternating
set -e
for i in 1 2; do
  echo "== current build"; python bench.py --gpus 1 --steps 20 --warmup 5 --main-only --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['sustained']['kernel_avg_ms'], d['roofline']['sustained']['kernel_min_ms'])"
  echo "== build of commit 2a25fa3"; BGS_LIB_PATH=$PWD/tracking_amd/lib/ab/libbgs_hip.so python bench.py --gpus 1 --steps 20 --warmup 5 --main-only --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['sustained']['kernel_avg_ms'], d['roofline']['sustained']['kernel_min_ms'])"
done
