"""Shared helpers of the GPU parity tests (not a test module): engine/oracle pairs and model-state comparisons."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import pyoracle
from tools import synth
from tracking_amd import Engine, capi


ALGOS = {
    "FrameDifferenceBGS": capi.FRAME_DIFF,
    "StaticFrameDifferenceBGS": capi.STATIC_FRAME_DIFF,
    "WeightedMovingMeanBGS": capi.WMM,
    "WeightedMovingVarianceBGS": capi.WMV,
    "AdaptiveBackgroundLearning": capi.ABL,
    "AdaptiveSelectiveBackgroundLearning": capi.ASBL,
    "MixtureOfGaussianV1BGS": capi.MOG1,
    "MixtureOfGaussianV2BGS": capi.MOG2,
    "SigmaDeltaBGS": capi.SIGMA_DELTA,
    "GMG": capi.GMG,
    "DPZivkovicAGMMBGS": capi.DP_ZIVKOVIC_AGMM,
    "DPGrimsonGMMBGS": capi.DP_GRIMSON_GMM,
    "DPWrenGABGS": capi.DP_WREN_GA,
    "DPMeanBGS": capi.DP_MEAN,
    "DPAdaptiveMedianBGS": capi.DP_ADAPTIVE_MEDIAN,
}


STATE_TOL = 1e-4


def max_err(a, b):
    """max |a - b| over the entries that are not identical; equal infinities and NaN on both sides count as identical (extreme
    parameters drive the reference's own arithmetic there, e.g. MOG2 with alpha 0.7 and cT 0.3 normalises by 1 / 0)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    with np.errstate(invalid="ignore"):
        d = np.abs(np.where(same, 0.0, a - b))
    return float(np.max(d)) if d.size else 0.0


def run_pair(algo, frames, params=None, want_bg=True, oparams=None):
    eng = Engine(algo, params=params)
    orc = pyoracle.Oracle(algo, params=oparams if oparams is not None else params)
    outs = []
    for t, f in enumerate(frames):
        fg, bg = eng.process(f, want_bg=want_bg)
        ofg, obg = orc.process(f, want_bg=want_bg)
        assert (fg is None) == (ofg is None), "frame %d: fg validity differs" % t
        assert (bg is None) == (obg is None), "frame %d: bg validity differs" % t
        if fg is not None:
            assert np.array_equal(fg, ofg), "frame %d: %d mask pixels differ" % (t, int((fg != ofg).sum()))
        if bg is not None:
            assert np.array_equal(bg, obg), "frame %d: %d background bytes differ" % (t, int((bg != obg).sum()))
        outs.append((fg, bg))
    return eng, orc, outs


def check_mog2_state(eng, orc, n, stream=0):
    for plane, shape, dt in (("w", (5, n), np.float32), ("var", (5, n), np.float32), ("mu", (5, 3, n), np.float32)):
        a, b = eng.get_state(plane, shape, dt, stream=stream), orc.get_state(plane, shape, dt)
        err = max_err(a, b)
        assert err <= STATE_TOL, "%s: max |delta| %g > %g" % (plane, err, STATE_TOL)
    assert np.array_equal(eng.get_state("nmodes", (n,), np.uint8, stream=stream), orc.get_state("nmodes", (n,), np.uint8))


def check_mog1_state(eng, orc, n, C=3, stream=0):
    for plane, shape in (("sortkey", (5, n)), ("w", (5, n)), ("mu", (5, C, n)), ("var", (5, C, n))):
        a, b = eng.get_state(plane, shape, np.float32, stream=stream), orc.get_state(plane, shape, np.float32)
        err = max_err(a, b)
        assert err <= STATE_TOL, "%s: max |delta| %g > %g" % (plane, err, STATE_TOL)


def check_dp_state(name, eng, orc, n, K=3, stream=0):
    """package_bgs/dp models: float planes within 1e-4 (observed 0), mode counts / median bytes exact."""
    planes = {"DPZivkovicAGMMBGS": ("modes", K * 5), "DPGrimsonGMMBGS": ("modes", K * 6), "DPWrenGABGS": ("gauss", 4), "DPMeanBGS": ("mean", 3)}
    if name in planes:
        plane, q = planes[name]
        a, b = eng.get_state(plane, (q, n), np.float32, stream=stream), orc.get_state(plane, (q, n), np.float32)
        err = max_err(a, b)
        assert err <= STATE_TOL, "%s %s: max |delta| %g" % (name, plane, err)
    if name in ("DPZivkovicAGMMBGS", "DPGrimsonGMMBGS"):
        assert np.array_equal(eng.get_state("nmodes", (n,), np.uint8, stream=stream), orc.get_state("nmodes", (n,), np.uint8))
    if name == "DPAdaptiveMedianBGS":
        assert np.array_equal(eng.get_state("median", (n * 3,), np.uint8, stream=stream), orc.get_state("median", (n * 3,), np.uint8))


def check_state(name, eng, orc, n, stream=0):
    if name.startswith("DP"):
        check_dp_state(name, eng, orc, n, stream=stream)
    if name == "MixtureOfGaussianV2BGS":
        check_mog2_state(eng, orc, n, stream)
    if name == "MixtureOfGaussianV1BGS":
        check_mog1_state(eng, orc, n, 3, stream)
    if name == "GMG":
        assert np.array_equal(eng.get_state("nfeatures", (n,), np.int32, stream=stream), orc.get_state("nfeatures", (n,), np.int32))
        assert np.array_equal(eng.get_state("colors", (64, n), np.int32, stream=stream), orc.get_state("colors", (64, n), np.int32))
        a, b = eng.get_state("weights", (64, n), np.float32, stream=stream), orc.get_state("weights", (64, n), np.float32)
        assert max_err(a, b) <= STATE_TOL
    if name == "SigmaDeltaBGS":
        for plane in ("mt", "vt"):
            assert np.array_equal(eng.get_state(plane, (n * 3,), np.uint8, stream=stream), orc.get_state(plane, (n * 3,), np.uint8)), plane
    if name in ("AdaptiveBackgroundLearning", "AdaptiveSelectiveBackgroundLearning", "StaticFrameDifferenceBGS"):
        c = 1 if name == "AdaptiveSelectiveBackgroundLearning" else 3
        assert np.array_equal(eng.get_state("bg", (n * c,), np.uint8, stream=stream), orc.get_state("bg", (n * c,), np.uint8))


def _params(algo, **kw):
    p = capi.default_params(algo)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


SS_F32 = ["R", "V", "T", "Dlast", "DminLT", "DminST", "RawLT", "RawST", "FinLT", "FinST"]
SS_U8 = ["unstable", "blinks", "lastfg", "lastraw"]


def check_subsense_state(eng, orc, rows, cols, nS=50, stream=0, C=3):
    n = rows * cols
    for pl in SS_F32:
        a, b = eng.get_state(pl, (n,), np.float32, stream=stream), orc.get_state(pl, (n,), np.float32)
        # inf/nan-free maps; tolerance as for every float state, observed difference 0
        assert np.array_equal(np.isfinite(a), np.isfinite(b)), pl
        err = float(np.max(np.abs(np.where(np.isfinite(a), a, 0) - np.where(np.isfinite(b), b, 0))))
        assert err <= STATE_TOL, "%s: max |delta| %g" % (pl, err)
    for pl in SS_U8:
        assert np.array_equal(eng.get_state(pl, (n,), np.uint8, stream=stream), orc.get_state(pl, (n,), np.uint8)), pl
    assert np.array_equal(eng.get_state("lastcolor", (n * C,), np.uint8, stream=stream), orc.get_state("lastcolor", (n * C,), np.uint8))
    assert np.array_equal(eng.get_state("lastdesc", (n * C,), np.uint16, stream=stream), orc.get_state("lastdesc", (n * C,), np.uint16))
    assert np.array_equal(eng.get_state("color", (nS, n, C), np.uint8, stream=stream), orc.get_state("color", (nS, n, C), np.uint8)), "colour samples"
    assert np.array_equal(eng.get_state("desc", (nS, n, C), np.uint16, stream=stream), orc.get_state("desc", (nS, n, C), np.uint16)), "descriptor samples"
    assert np.array_equal(eng.get_state("lut", (256,), np.uint8, stream=stream), orc.get_state("lut", (256,), np.uint8))
    assert np.array_equal(eng.get_state("scalars", (7,), np.float64, stream=stream), orc.get_state("scalars", (7,), np.float64))


def _cc_masks(shape, rng):
    rows, cols = shape
    yield "empty", np.zeros(shape, np.uint8)
    yield "full", np.full(shape, 255, np.uint8)
    for density in (0.05, 0.3, 0.45, 0.6):  # 0.45-0.6: around the percolation thresholds of 8- and 4-connectivity
        yield "random%.2f" % density, np.where(rng.random(shape) < density, 255, 0).astype(np.uint8)
    m = np.zeros(shape, np.uint8)  # isolated pixels on a lattice: the maximum number of components
    m[::2, ::2] = 1
    yield "lattice", m
    m = np.zeros(shape, np.uint8)  # diagonal staircases: joined only under 8-connectivity
    for k in range(0, rows + cols, 7):
        for t in range(min(rows, cols)):
            y, x = t, k - t
            if 0 <= x < cols:
                m[y, x] = 200
    yield "diagonals", m
    if rows > 8 and cols > 8:
        m = np.zeros(shape, np.uint8)  # one serpentine component spanning the image: long union-find chains
        m[1:-1:4, 1:-1] = 255
        m[1:-1, 1] = 255
        m[3:-1:8, 1] = 0
        m[1:-1, -2] |= np.where((np.arange(rows - 2) // 4) % 2 == 0, 255, 0).astype(np.uint8)
        yield "serpentine", m
        m = np.zeros(shape, np.uint8)  # blobs like a foreground mask: filled rectangles and rings
        for _ in range(12):
            y, x = rng.integers(0, rows - 4), rng.integers(0, cols - 4)
            h, w = rng.integers(2, max(3, rows // 3)), rng.integers(2, max(3, cols // 3))
            m[y:y + h, x:x + w] = 255
            if h > 6 and w > 6:
                m[y + 2:y + h - 2, x + 2:x + w - 2] = 0
        yield "blobs", m


DP_NAMES = ["DPZivkovicAGMMBGS", "DPGrimsonGMMBGS", "DPWrenGABGS", "DPMeanBGS", "DPAdaptiveMedianBGS"]


def check_lobster_state(eng, orc, rows, cols, nS=35, stream=0, C=3):
    n = rows * cols
    assert np.array_equal(eng.get_state("lastfg", (n,), np.uint8, stream=stream), orc.get_state("lastfg", (n,), np.uint8))
    assert np.array_equal(eng.get_state("lastcolor", (n * C,), np.uint8, stream=stream), orc.get_state("lastcolor", (n * C,), np.uint8))
    assert np.array_equal(eng.get_state("lastdesc", (n * C,), np.uint16, stream=stream), orc.get_state("lastdesc", (n * C,), np.uint16))
    assert np.array_equal(eng.get_state("color", (nS, n, C), np.uint8, stream=stream), orc.get_state("color", (nS, n, C), np.uint8)), "colour samples"
    assert np.array_equal(eng.get_state("desc", (nS, n, C), np.uint16, stream=stream), orc.get_state("desc", (nS, n, C), np.uint16)), "descriptor samples"
    assert np.array_equal(eng.get_state("lut", (256,), np.uint8, stream=stream), orc.get_state("lut", (256,), np.uint8))
