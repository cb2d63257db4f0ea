"""Diagnostic (round 3): where does the time of bgs_submit / bgs_wait go with 1, 2, 4, 8 cameras, staged and with registered buffers."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from tools import synth
from tracking_amd import Engine, capi

ROWS, COLS = 1080, 1920
clip = synth.s_sat(10, ROWS, COLS, seed=1234, device="cuda").cpu().numpy()
for cams in (1, 2, 4, 8):
    for reg in (0, 3):
        e = Engine(capi.MOG2, n_streams=cams)
        e.set_option(capi.OPT_HOST_REGISTER, reg)
        frames = [np.ascontiguousarray(clip[c % 10]) for c in range(cams)]
        fgs = [np.empty((ROWS, COLS), np.uint8) for _ in range(cams)]
        for t in range(10):
            for c in range(cams):
                e.submit(frames[c], fgs[c], None, stream=c)
            for c in range(cams):
                e.wait(stream=c)
        rounds = 30
        ts = tw = 0.0
        t0 = time.perf_counter()
        for t in range(rounds):
            a = time.perf_counter()
            for c in range(cams):
                e.submit(frames[c], fgs[c], None, stream=c)
            b = time.perf_counter()
            for c in range(cams):
                e.wait(stream=c)
            ts += b - a
            tw += time.perf_counter() - b
        dt = time.perf_counter() - t0
        # the same cameras through the synchronous call, one after the other
        t1 = time.perf_counter()
        for t in range(rounds):
            for c in range(cams):
                e.process_into(frames[c], fgs[c], None, stream=c)
        ds = time.perf_counter() - t1
        print("cams %d reg %d: submit/wait %.3f ms per frame (submit part %.3f, wait part %.3f); synchronous calls %.3f ms per frame" % (cams, reg, dt / rounds / cams * 1e3, ts / rounds / cams * 1e3, tw / rounds / cams * 1e3, ds / rounds / cams * 1e3), flush=True)
        e.close()
