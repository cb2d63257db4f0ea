#!/usr/bin/env python3
"""SuBSENSE, 8 x 1080p per step on fresh-noise frames (the driver line's block): the step as ONE batch call, and as `groups` stream
ranges on HIP streams of their own (bgs_process_range_device) - with the phase A token (engine_subsense.h) the ranges' phase A
launches take turns and every range's tail runs beside another range's phase A.  Young model (6 frames, a fresh engine per leg) and
aged model (300 frames, one engine, legs alternating).  Knobs are read once per process: run it once per setting
(tools/r04_ss_token.sh).
usage: r04_ss_token.py [--groups 1,2,4] [--young 1] [--aged 1]"""
import argparse
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # one hardware queue per HIP stream in use (4 by default: streams beyond share them)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools import synth
from tracking_amd import Engine, capi

ROWS, COLS, S, T = 1080, 1920, 8, 36


_HS = []


def stepper(e, fg, groups):
    while len(_HS) < groups:  # (torch hands out streams from a pool and never destroys them: take each one once)
        _HS.append(torch.cuda.Stream())
    hs = _HS
    per = S // groups

    def step(frames):
        if groups == 1:
            e.process_batch_device(frames, fg, None, None)
        else:
            for g in range(groups):
                e.process_batch_device(frames[g * per:(g + 1) * per], fg[g * per:(g + 1) * per], None, None, hip_stream=hs[g].cuda_stream, first=g * per, count=per)
    return step


def timed(e, step, pool, t0, n):
    torch.cuda.synchronize()
    e.enable_kernel_timing(True)
    w0 = time.perf_counter()
    for i in range(n):
        step(pool[(t0 + i) % T])
    torch.cuda.synchronize()
    wall = (time.perf_counter() - w0) / n * 1e3
    ms, launches, _ = e.kernel_timing()
    e.enable_kernel_timing(False)
    return wall, ms * launches / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", default="1,2,4")
    ap.add_argument("--young", type=int, default=1)
    ap.add_argument("--aged", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    groups = [int(g) for g in args.groups.split(",")]
    dev = torch.device("cuda", 0)
    knobs = {k: os.environ[k] for k in ("BGS_SS_A_TOKEN", "BGS_SS_PARTS", "GPU_MAX_HW_QUEUES", "BGS_SS_B_LATE") if k in os.environ}
    print("knobs", knobs, flush=True)
    src = synth.SurvStreams(S, ROWS, COLS, seed0=4321, device=dev)
    pool = src.pool(T)
    fg = torch.empty((S, ROWS, COLS), dtype=torch.uint8, device=dev)
    if args.young:
        for g in groups:
            e = Engine(capi.SUBSENSE, n_streams=S)
            e.set_geometry(ROWS, COLS, 3)
            step = stepper(e, fg, g)
            e.process_batch_device(pool[0], fg, None, None)
            torch.cuda.synchronize()
            for t in range(1, 6):
                step(pool[t])
            wall, a_ms = timed(e, step, pool, 6, args.steps)
            print("young (age 6)   groups %d: %.3f ms per 8 x 1080p step; phase A %.3f ms per step; fg %.4f" % (g, wall, a_ms, float((fg != 0).float().mean())), flush=True)
            e.close()
            del e
    if args.aged:
        e = Engine(capi.SUBSENSE, n_streams=S)
        e.set_geometry(ROWS, COLS, 3)
        cur = torch.empty((S, ROWS, COLS, 3), dtype=torch.uint8, device=dev)
        e.process_batch_device(pool[0], fg, None, None)
        for t in range(1, 300):
            e.process_batch_device(src.into(cur), fg, None, None)
        for rep in range(2):
            for g in groups:
                pool[:args.steps] = src.pool(args.steps)  # frames the model has never seen
                step = stepper(e, fg, g)
                for t in range(3):
                    f = src.into(cur)
                    torch.cuda.synchronize()  # (the ranges' streams do not follow torch's current stream)
                    step(f)
                    torch.cuda.synchronize()
                wall, a_ms = timed(e, step, pool, 0, args.steps)
                print("aged (age 300+) groups %d: %.3f ms per 8 x 1080p step; phase A %.3f ms per step; fg %.4f" % (g, wall, a_ms, float((fg != 0).float().mean())), flush=True)
        e.close()


if __name__ == "__main__":
    main()
