#!/usr/bin/env python3
"""MOG2 on S_surv, 32 x 1080p: the automatic kernel choice (BGS_DEBUG_STAT=1 prints every decision) beside each fixed mode."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

S = 32
fg = torch.empty((S, bench.ROWS, bench.COLS), dtype=torch.uint8, device="cuda:0")
for sparse in (3, 1, 2, 4):
    r = bench.surv_leg(0, S, fg, sparse=sparse)
    print("sparse %d: kernel %.4f ms, %.1f Mpix/s wall" % (sparse, r["kernel_ms"], r["mpixels_per_s"]), flush=True)
