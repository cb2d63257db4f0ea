#!/bin/bash
# The MOG2 headline kernel, alternating on one box:
#   current        this tree (filter kernel without the shadow / background code: 66 VGPRs = 7 waves per SIMD)
#   current, pad   the same held to 6 / 5 workgroups per CU by unused dynamic LDS (BGS_MOG2_LDS_PAD)
#   ab0            the tree of the commit before (91 VGPRs = 5 waves per SIMD)
one() { python bench.py --gpus 1 --steps 20 --warmup 5 --main-only --no-pmc --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('timed(fresh) %.4f  sustained %.4f  min %.4f' % (r['kernel_avg_ms'], r['sustained']['kernel_avg_ms'], r['sustained']['kernel_min_ms']))"; }
for i in 1 2 3 4; do
  echo -n "current (7 waves)   "; one
  echo -n "current, 6 waves    "; BGS_MOG2_LDS_PAD=23000 one
  echo -n "current, 5 waves    "; BGS_MOG2_LDS_PAD=27000 one
  [ -f tracking_amd/lib/ab0/libbgs_hip.so ] && { echo -n "ab0 (5 waves)       "; BGS_LIB_PATH=$PWD/tracking_amd/lib/ab0/libbgs_hip.so one; }
done
