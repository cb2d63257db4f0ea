#!/usr/bin/env python3
"""SuBSENSE model initialisation, 8 x 1080p: every stream reset and the first frame again, three times (for a kernel trace:
tools/prof_any.sh init tools/r04_init.py -> ss_refresh_kernel, ss_lastrec_pack_kernel, lbsp_kernel, the fills)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools import synth
from tracking_amd import Engine, capi

S, ROWS, COLS = 8, 1080, 1920
dev = torch.device("cuda", 0)
src = synth.SurvStreams(S, ROWS, COLS, seed0=4321, device=dev)
pool = src.pool(2)
fg = torch.empty((S, ROWS, COLS), dtype=torch.uint8, device=dev)
e = Engine(capi.SUBSENSE, n_streams=S)
e.set_geometry(ROWS, COLS, 3)
for rep in range(4):
    for s in range(S):
        e.reset_stream(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e.process_batch_device(pool[0], fg, None, None)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    e.process_batch_device(pool[1], fg, None, None)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("rep %d: initialising call %.3f ms, the step behind it %.3f ms" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
e.close()
