#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ (run in the build container, where /root/reference exists).

Inputs  : crops of the reference's own test data files  /root/reference/frames/1..51.png  (RGB PNG 320x240;
          converted RGB->BGR the way cv::imread hands them to IBGS::process).  Data only — no reference source.
Outputs :
  frames_96x80.npz      'frames' uint8 [24][80][96][3]  (BGR, frames 1..24, crop y 60:140, x 110:206 — the busiest region)
  frames_gray_64x48.npz 'frames' uint8 [12][48][64]
  sigmadelta_ref.npz    SigmaDeltaBGS masks of frames[1:] produced by THE REFERENCE'S OWN package_bgs/bl/sdLaMa091.cpp
                        (oracle/_ref/ref_sdlama_cli), default parameters and (ampFactor 3, minVar 2, maxVar 200) -> pinned
  lbsp_ref.npz          LBSP descriptors of frames[0] produced by THE REFERENCE'S OWN pattern files
                        (oracle/_ref/libref_lbsp.so, built from package_bgs/pl/LBSP_16bits_dbcross_*.i) -> pinned
  framediff_indep.npz   FrameDifference masks computed by an independent numpy formula (integer only)
  oracle_regress.npz    masks / backgrounds / MOG2 state from the CPU oracle for every algorithm.  These pin the
                        oracle against accidental change; they are NOT reference outputs (OpenCV absent -> unpinned).
"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from tracking_amd import capi  # noqa: E402

REF = "/root/reference/frames"


def load(i):
    return np.ascontiguousarray(np.array(Image.open(os.path.join(REF, "%d.png" % i)).convert("RGB"))[:, :, ::-1])


def main():
    frames = np.stack([load(i)[60:140, 110:206] for i in range(1, 25)])
    np.savez_compressed(os.path.join(HERE, "frames_96x80.npz"), frames=frames)
    gray = np.stack([po.bgr2gray(load(i))[100:148, 150:214] for i in range(1, 13)])
    np.savez_compressed(os.path.join(HERE, "frames_gray_64x48.npz"), frames=gray)

    # --- pinned: reference-built LBSP
    assert po.ref_lbsp_available(), "oracle/_ref not built (needs /root/reference)"
    lut3, lut1 = po.lbsp_lut(0.333, 0, 3), po.lbsp_lut(0.333, 0, 1)
    np.savez_compressed(os.path.join(HERE, "lbsp_ref.npz"), lut3=lut3, lut1=lut1,
                        desc3=po.ref_lbsp_describe(frames[0], lut3), desc1=po.ref_lbsp_describe(gray[0], lut1)[:, :, 0],
                        desc3_f7=po.ref_lbsp_describe(frames[7], lut3))

    # --- pinned: SigmaDelta masks from the reference's own sdLaMa091.cpp (oracle/_ref/ref_sdlama_cli)
    assert po.ref_sdlama_available()
    sd = {}
    for tag, (amp, vmin, vmax) in {"default": (1, 15, 255), "amp3": (3, 2, 200)}.items():
        sd[tag] = po.ref_sigmadelta_clip(frames, amp, vmin, vmax)
    np.savez_compressed(os.path.join(HERE, "sigmadelta_ref.npz"), **sd)

    # --- independent integer formula for FrameDifference (FrameDifferenceBGS.cpp:45-51 with P1/P2 of DESIGN.md §5)
    masks = []
    for a, b in zip(frames[:-1], frames[1:]):
        d = np.abs(a.astype(np.int32) - b.astype(np.int32))
        g = (d[:, :, 0] * 1868 + d[:, :, 1] * 9617 + d[:, :, 2] * 4899 + 8192) >> 14
        masks.append(np.where(g > 15, 255, 0).astype(np.uint8))
    np.savez_compressed(os.path.join(HERE, "framediff_indep.npz"), masks=np.stack(masks))

    # --- oracle regression vectors
    out = {}
    for name, algo in [("fd", capi.FRAME_DIFF), ("sfd", capi.STATIC_FRAME_DIFF), ("wmm", capi.WMM), ("wmv", capi.WMV),
                       ("abl", capi.ABL), ("asbl", capi.ASBL), ("mog2", capi.MOG2), ("mog1", capi.MOG1), ("sd", capi.SIGMA_DELTA), ("gmg", capi.GMG), ("subsense", capi.SUBSENSE),
                       ("dpziv", capi.DP_ZIVKOVIC_AGMM), ("dpgrim", capi.DP_GRIMSON_GMM), ("dpwren", capi.DP_WREN_GA), ("dpmean", capi.DP_MEAN), ("dpmedian", capi.DP_ADAPTIVE_MEDIAN), ("lobster", capi.LOBSTER)]:
        o = po.Oracle(algo)
        fgs, bgs = [], []
        for f in frames:
            fg, bg = o.process(f)
            fgs.append(fg if fg is not None else np.full(f.shape[:2], 7, np.uint8))  # 7 = "output untouched" marker
            if bg is not None:
                bgs.append(bg)
        out[name + "_fg"] = np.stack(fgs)
        if bgs:
            out[name + "_bg_last"] = bgs[-1]
        if name == "mog2":
            n = frames.shape[1] * frames.shape[2]
            out["mog2_w"] = o.get_state("w", (5, n), np.float32)
            out["mog2_var"] = o.get_state("var", (5, n), np.float32)
            out["mog2_mu"] = o.get_state("mu", (5, 3, n), np.float32)
            out["mog2_nmodes"] = o.get_state("nmodes", (n,), np.uint8)
    np.savez_compressed(os.path.join(HERE, "oracle_regress.npz"), **out)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
