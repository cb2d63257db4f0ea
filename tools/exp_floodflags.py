import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from tools import synth
from tracking_amd import Engine, capi
S, rows, cols, T = 4, 1080, 1920, 8
for kind in ("surv", "smooth"):
    pool = torch.empty((T, S, rows, cols, 3), dtype=torch.uint8, device="cuda")
    for s in range(S):
        pool[:, s] = (synth.s_surv if kind == "surv" else synth.s_smooth)(T, rows, cols, seed=4321 + s, device="cuda")
    e = Engine(capi.SUBSENSE, n_streams=S); e.set_geometry(rows, cols, 3)
    fg = torch.empty((S, rows, cols), dtype=torch.uint8, device="cuda")
    for t in range(14):
        e.process_batch_device(pool[t % T], fg, None, None)
        if t in (0, 5, 13):
            print(kind, "frame", t, [e.get_state("floodflags", (9,), np.int32, stream=s).tolist() for s in range(2)])
    e.close()
