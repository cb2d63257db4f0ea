#!/usr/bin/env python3
"""Interleaved A/B of MOG2 kernel variants in ONE process (same device, same clocks): round-robin over the variants,
`reps` rounds of `steps` launches each, report the median algorithmic GB/s per variant.
usage: ab_mog2.py "layout:px[:env=val,...]" ...   e.g.  ab_mog2.py planar:4 tiled:4 planar:1"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tracking_amd import Engine, capi
from tools import synth

ROWS, COLS, S = 1080, 1920, int(os.environ.get("AB_STREAMS", "32"))
STEPS, REPS = int(os.environ.get("AB_STEPS", "40")), int(os.environ.get("AB_REPS", "7"))


def main():
    variants = sys.argv[1:] or ["planar:4", "tiled:4"]
    dev = torch.device("cuda", 0)
    print(torch.cuda.get_device_name(0), "mem free/total GiB", [round(x / 2**30, 1) for x in torch.cuda.mem_get_info()])
    pad_gib = float(os.environ.get("AB_PAD_GIB", "0"))
    pad = torch.empty(int(pad_gib * 2**30), dtype=torch.uint8, device=dev) if pad_gib else None
    period = 10
    pool = torch.empty((period, S, ROWS, COLS, 3), dtype=torch.uint8, device=dev)
    for s in range(S):
        pool[:, s] = synth.s_sat(period, ROWS, COLS, seed=1234 + s, device=dev)
    fg = torch.empty((S, ROWS, COLS), dtype=torch.uint8, device=dev)
    engines = []
    variants = ["%s#%d" % (v, i) for i, v in enumerate(variants)]
    for v in variants:
        parts = v.split("#")[0].split(":")
        layout, px = parts[0], int(parts[1])
        envs = [kv.split("=") for kv in (parts[2].split(",") if len(parts) > 2 else [])]
        for k, val in envs:
            os.environ["BGS_" + k] = val
        e = Engine(capi.MOG2, n_streams=S)
        for k, _ in envs:
            del os.environ["BGS_" + k]
        e.set_option(capi.OPT_MOG2_TILED, 1 if layout == "tiled" else 0)
        e.set_option(capi.OPT_MOG2_PIXELS_PER_LANE, px)
        e.set_geometry(ROWS, COLS, 3)
        for t in range(50):
            e.process_batch_device(pool[t % period], fg, None, None)
        engines.append((v, e, [50]))
    torch.cuda.synchronize()
    res = {v: [] for v in variants}
    import random
    random.seed(5)
    for r in range(REPS):
        order = list(engines)
        if os.environ.get("AB_SHUFFLE"):
            random.shuffle(order)
            print("rep", r, "order", [v.split("#")[1] for v, _, _ in order])
        for v, e, tt in order:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(STEPS):
                e.process_batch_device(pool[tt[0] % period], fg, None, None)
                tt[0] += 1
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[v].append(206.0 * S * ROWS * COLS * STEPS / dt / 1e9)
    for v in variants:
        a = np.array(res[v])
        print("%-28s median %7.1f GB/s  min %7.1f  max %7.1f   (%s)" % (v, np.median(a), a.min(), a.max(), " ".join("%.0f" % x for x in a)))


if __name__ == "__main__":
    main()
