#!/bin/bash
# MOG2 tile size A/B (pixels per tile: 256 = the tree's, 128 / 512 / 1024 = builds with -DBGS_MOG2_TILE), same box, alternating
one() { python bench.py --gpus 1 --steps 20 --warmup 5 --main-only --no-pmc --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('timed(fresh) %.4f  sustained %.4f  min %.4f' % (r['kernel_avg_ms'], r['sustained']['kernel_avg_ms'], r['sustained']['kernel_min_ms']))"; }
for i in 1 2; do
  echo -n "tile 256   "; one
  for T in 128 512 1024; do echo -n "tile $T   "; BGS_LIB_PATH=$PWD/tracking_amd/lib/abt$T/libbgs_hip.so one; done
done
