#!/bin/bash
# The MOG2 headline kernel (fresh frames: timed; the same 25 frames repeating: sustained), alternating on one box:
#   current   this tree
#   ab1       tracking_amd/lib/ab1  (the same tree before the 2-byte summaries: lock-step update, 4-byte summaries)
#   ab0       tracking_amd/lib/ab0  (round 3's kernel: the library of this round's first commit)
one() { python bench.py --gpus 1 --steps 20 --warmup 5 --main-only --no-pmc --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('timed(fresh) %.4f  sustained %.4f  min %.4f' % (r['kernel_avg_ms'], r['sustained']['kernel_avg_ms'], r['sustained']['kernel_min_ms']))"; }
for i in 1 2 3; do
  echo -n "current   "; one
  for v in ab1 ab0; do
    [ -f tracking_amd/lib/$v/libbgs_hip.so ] && { echo -n "$v       "; BGS_LIB_PATH=$PWD/tracking_amd/lib/$v/libbgs_hip.so BGS_LIB_PARTIAL_ABI=1 one; }
  done
done
