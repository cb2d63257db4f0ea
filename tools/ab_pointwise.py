"""Interleaved A/B of the byte-stream kernels: pixels per lane (BGS_FRAME_GROUP 4 / 16) x XCD-aware block order (on / off),
same input buffers, same process, 3 rounds.  8 x 3840x2160 S_surv streams."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import synth  # noqa: E402
from tracking_amd import Engine, capi  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    S, rows, cols, T = 8, 2160, 3840, 8
    pool = torch.empty((T, S, rows, cols, 3), dtype=torch.uint8, device=dev)
    for s in range(S):
        pool[:, s] = synth.s_surv(T, rows, cols, seed=4321 + s, device=dev)
    fg = torch.empty((S, rows, cols), dtype=torch.uint8, device=dev)
    for algo, name, borrow in ((capi.FRAME_DIFF, "framediff", True), (capi.WMV, "wmv", True), (capi.WMM, "wmm", True), (capi.ABL, "abl", False), (capi.SIGMA_DELTA, "sigmadelta", False)):
        res = {}
        for rnd in range(3):
            for g in (16, 4):
                for swz in (1, 0):
                    os.environ["BGS_FRAME_GROUP"] = str(g)
                    e = Engine(algo, n_streams=S)
                    e.set_geometry(rows, cols, 3)
                    e.set_option(capi.OPT_XCD_SWIZZLE, swz)
                    if borrow:
                        e.set_option(capi.OPT_BORROW_FRAMES, 1)
                    for t in range(6):
                        e.process_batch_device(pool[t % T], fg, None, None)
                    torch.cuda.synchronize()
                    e.enable_kernel_timing(True)
                    for t in range(40):
                        e.process_batch_device(pool[(6 + t) % T], fg, None, None)
                    torch.cuda.synchronize()
                    ms, n, _ = e.kernel_timing()
                    res.setdefault((g, swz), []).append(ms)
                    e.close()
        print(name, "  ".join("G%d/swz%d: %s" % (g, swz, " ".join("%.4f" % v for v in res[(g, swz)])) for (g, swz) in sorted(res)))


if __name__ == "__main__":
    main()
