"""MixtureOfGaussianV1BGS update kernel on frames with FRESH sensor noise (tools/synth.py SurvStreams / SatStreams): 16 x 1080p, the model aged on
frames generated one by one, the timed launches over a pool of distinct frames.  (tools/bench_configs.py's MOG1 legs cycle a pool of 8 - 10 frames.)
A/B: BGS_LIB_PATH=<other build> python tools/r04_mog1_fresh.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synth  # noqa: E402
from tracking_amd import capi  # noqa: E402
from tracking_amd.engine import Engine  # noqa: E402


def leg(kind, S=16, rows=1080, cols=1920, warm=120, steps=30):
    dev = torch.device("cuda", 0)
    src = (synth.SurvStreams if kind == "surv" else synth.SatStreams)(S, rows, cols, seed0=4321, device=dev)
    e = Engine(capi.MOG1, n_streams=S)
    e.set_geometry(rows, cols, 3)
    fg = torch.empty((S, rows, cols), dtype=torch.uint8, device=dev)
    cur = torch.empty((S, rows, cols, 3), dtype=torch.uint8, device=dev)
    for _ in range(warm):
        e.process_batch_device(src.into(cur), fg, None, None)
    pool = src.pool(steps)
    torch.cuda.synchronize()
    e.enable_kernel_timing(True)
    for t in range(steps):
        e.process_batch_device(pool[t], fg, None, None)
    torch.cuda.synchronize()
    ms, n, kname = e.kernel_timing()
    px = S * rows * cols
    print("MOG1 %-5s fresh noise, %d x %dx%d: %s %.4f ms -> %.1f Gpixel/s; foreground %.4f" % (kind, S, cols, rows, kname, ms, px / ms / 1e6, float((fg != 0).float().mean())))
    e.close()
    del pool, cur, fg
    torch.cuda.empty_cache()


if __name__ == "__main__":
    leg("surv")
    leg("sat")
