"""Counting build of phase A (tracking_amd/lib/exp/lib_COUNT.so, -DBGS_EXP_COUNT): wave-iterations of the sample loop, active lane-trips,
wave-iterations that run the inter-LBSP part, lanes that need it.  usage: BGS_LIB_PATH=.../lib_COUNT.so python tools/exp_count.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synth
from tracking_amd import Engine, capi

S, rows, cols, T = 8, 1080, 1920, 8
pool = torch.empty((T, S, rows, cols, 3), dtype=torch.uint8, device="cuda")
for s in range(S):
    pool[:, s] = synth.s_surv(T, rows, cols, seed=4321 + s, device="cuda")
e = Engine(capi.SUBSENSE, n_streams=S)
e.set_geometry(rows, cols, 3)
fg = torch.empty((S, rows, cols), dtype=torch.uint8, device="cuda")
for t in range(12):
    e.process_batch_device(pool[t % T], fg, None, None)
torch.cuda.synchronize()
lib = C.CDLL(capi.LIB_PATH)
out = (C.c_ulonglong * 8)()
lib.bgs_debug_counters(out, 1)
n = 10
for t in range(n):
    e.process_batch_device(pool[(12 + t) % T], fg, None, None)
torch.cuda.synchronize()
lib.bgs_debug_counters(out, 1)
px = S * rows * cols * n
print("refills per pixel-round (x64/px) %.2f, lanes refilled per refill %.1f" % (out[4] * 64 / px, out[5] / max(out[4], 1)))
print("per pixel: wave-iterations x64 %.2f  active lane-trips %.2f  (utilisation %.2f)  inter-LBSP wave-iterations x64 %.2f  lanes needing it %.2f"
      % (out[0] * 64 / px, out[1] / px, out[1] / (out[0] * 64.0), out[2] * 64 / px, out[3] / px))
