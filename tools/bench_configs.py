#!/usr/bin/env python3
"""Secondary workloads of BASELINE.json (configs[2], configs[3]) and the other byte kernels, kernel time by HIP events.
Not the driver's bench line (bench.py is); the numbers go to DESIGN.md §6.
  configs[2]: WeightedMovingVarianceBGS + AdaptiveBackgroundLearning, 3840x2160 (HBM-bound stress)
  configs[3]: LBSP descriptor path, 1920x1080
usage: bench_configs.py [--streams S] [--swizzle 0|1]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tools import synth
from tracking_amd import Engine, capi
from tracking_amd.engine import lbsp_describe_device


def run(algo, name, rows, cols, S, bpp, steps=60, borrow=True, want_bg=False):
    dev = torch.device("cuda", 0)
    T = 8
    pool = torch.empty((T, S, rows, cols, 3), dtype=torch.uint8, device=dev)
    for s in range(S):
        pool[:, s] = synth.s_surv(T, rows, cols, seed=4321 + s, device=dev)
    e = Engine(algo, n_streams=S)
    e.set_geometry(rows, cols, 3)
    if borrow:
        e.set_option(capi.OPT_BORROW_FRAMES, 1)
    fg = torch.empty((S, rows, cols), dtype=torch.uint8, device=dev)
    bg = torch.empty((S, rows, cols, 3), dtype=torch.uint8, device=dev) if want_bg else None
    for t in range(10):
        e.process_batch_device(pool[t % T], fg, bg, None)
    torch.cuda.synchronize()
    e.enable_kernel_timing(True)
    t0 = time.perf_counter()
    for t in range(steps):
        e.process_batch_device(pool[(10 + t) % T], fg, bg, None)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms, n, kname = e.kernel_timing()
    px = S * rows * cols
    print("%-34s %dx%d x%d streams: kernel %-18s %.4f ms  -> %8.1f Mpix/s  %7.1f GB/s algorithmic (%d B/px) = %.1f%% of 8 TB/s | wall %.1f Mpix/s"
          % (name, cols, rows, S, kname, ms, px / ms / 1e3, bpp * px / ms / 1e6, bpp, bpp * px / ms / 1e6 / 80.0, px * steps / wall / 1e6))
    e.close()


def run_subsense(S, steps=30, kind="surv"):
    """BASELINE configs[3]: SuBSENSE at 1920x1080 (per-frame wall time: ~20 launches incl. the flood-fill host loop)."""
    dev = torch.device("cuda", 0)
    rows, cols, T = 1080, 1920, 8
    pool = torch.empty((T, S, rows, cols, 3), dtype=torch.uint8, device=dev)
    for s in range(S):
        pool[:, s] = (synth.s_surv if kind == "surv" else synth.s_smooth)(T, rows, cols, seed=4321 + s, device=dev)
    e = Engine(capi.SUBSENSE, n_streams=S)
    e.set_geometry(rows, cols, 3)
    fg = torch.empty((S, rows, cols), dtype=torch.uint8, device=dev)
    for t in range(6):
        e.process_batch_device(pool[t % T], fg, None, None)
    torch.cuda.synchronize()
    e.enable_kernel_timing(True)
    t0 = time.perf_counter()
    for t in range(steps):
        e.process_batch_device(pool[(6 + t) % T], fg, None, None)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    ms, n, kname = e.kernel_timing()
    px = S * rows * cols
    print("%-34s %dx%d x%d streams: %.3f ms/frame-step wall -> %8.1f Mpix/s (%.1f 1080p frames/s); %s %.3f ms; fg ratio %.3f"
          % ("SuBSENSEBGS (%s input)" % kind, cols, rows, S, wall * 1e3, px / wall / 1e6, S / wall, kname, ms, float((fg != 0).float().mean())))
    e.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=8)
    args = ap.parse_args()
    S = args.streams
    run(capi.WMV, "WeightedMovingVarianceBGS", 2160, 3840, S, 10)
    run(capi.ABL, "AdaptiveBackgroundLearning", 2160, 3840, S, 10, borrow=False)
    run(capi.WMM, "WeightedMovingMeanBGS (+bg)", 2160, 3840, S, 13, want_bg=True)
    run(capi.FRAME_DIFF, "FrameDifferenceBGS", 2160, 3840, S, 7)
    run(capi.MOG1, "MixtureOfGaussianV1BGS", 1080, 1920, 16, 324, borrow=False)
    run_subsense(2)
    run_subsense(2, kind="smooth")
    # LBSP descriptors, 1080p
    img = synth.s_surv(1, 1080, 1920, seed=9, device="cuda")[0]
    from oracle import pyoracle
    lut = pyoracle.lbsp_lut(0.333, 0, 3)
    out = torch.empty((1080, 1920, 3), dtype=torch.int16, device="cuda")
    for _ in range(5):
        lbsp_describe_device(img, lut, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        lbsp_describe_device(img, lut, out=out)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 50
    px = 1080 * 1920
    print("%-34s 1920x1080 x1: %.4f ms -> %8.1f Mpix/s  %7.1f GB/s algorithmic (9 B/px)" % ("LBSP descriptors (lbsp_kernel)", ms, px / ms / 1e3, 9 * px / ms / 1e6))


if __name__ == "__main__":
    main()
