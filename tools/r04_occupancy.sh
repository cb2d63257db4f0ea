#!/bin/bash
# How much does the MOG2 headline kernel depend on the waves resident per SIMD?  Unused dynamic LDS per workgroup (BGS_MOG2_LDS_PAD) lets
# 5 (default: 91 VGPRs), 4, 3, 2 workgroups = waves per SIMD fit a CU.
one() { python bench.py --gpus 1 --steps 20 --warmup 5 --main-only --no-pmc --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('timed(fresh) %.4f  sustained %.4f  min %.4f' % (r['kernel_avg_ms'], r['sustained']['kernel_avg_ms'], r['sustained']['kernel_min_ms']))"; }
for pad in 0 33000 41000 54000 0; do
  echo -n "lds pad $pad  "; BGS_MOG2_LDS_PAD=$pad one
done
