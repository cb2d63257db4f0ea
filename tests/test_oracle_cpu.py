"""CPU tests of the oracle itself (no GPU): golden vectors, the reference-built checkers, the reference's quirks.

What pins what:
  * LBSP descriptors   — fixture produced by the REFERENCE'S OWN pattern files (oracle/_ref) -> pinned
  * FrameDifference    — independent integer formula (numpy) written from FrameDifferenceBGS.cpp:45-51
  * everything else    — regression vectors of the oracle (OpenCV is absent: parity unpinned), plus properties
"""
import os

import numpy as np
import pytest

from oracle import pyoracle
from tools import synth
from tracking_amd import capi

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
UNTOUCHED = 7  # marker make_golden.py stores where the reference leaves the output untouched


def test_lbsp_oracle_matches_reference_built_fixture(golden_frames, golden_gray):
    g = np.load(os.path.join(GOLDEN, "lbsp_ref.npz"))
    assert np.array_equal(pyoracle.lbsp_lut(0.333, 0, 3), g["lut3"])
    assert np.array_equal(pyoracle.lbsp_lut(0.333, 0, 1), g["lut1"])
    assert np.array_equal(pyoracle.lbsp_describe(golden_frames[0], g["lut3"]), g["desc3"])
    assert np.array_equal(pyoracle.lbsp_describe(golden_frames[7], g["lut3"]), g["desc3_f7"])
    assert np.array_equal(pyoracle.lbsp_describe(golden_gray[0], g["lut1"])[:, :, 0], g["desc1"])


@pytest.mark.skipif(not pyoracle.ref_lbsp_available(), reason="oracle/_ref is built only where /root/reference exists")
def test_lbsp_oracle_matches_live_reference_build():
    rng = np.random.default_rng(1)
    for ch in (1, 3):
        img = rng.integers(0, 256, (40, 56) + ((3,) if ch == 3 else ()), dtype=np.uint8)
        for rel, off in ((0.333, 0), (0.05, 9), (1.0, 0)):
            lut = pyoracle.lbsp_lut(rel, off, ch)
            assert np.array_equal(pyoracle.lbsp_describe(img, lut), pyoracle.ref_lbsp_describe(img, lut))


def test_sigmadelta_oracle_matches_reference_built_fixture(golden_frames):
    """tests/golden/sigmadelta_ref.npz = masks from the reference's own sdLaMa091.cpp (made in the build container)."""
    g = np.load(os.path.join(GOLDEN, "sigmadelta_ref.npz"))
    for tag, (amp, vmin, vmax) in {"default": (1, 15, 255), "amp3": (3, 2, 200)}.items():
        p = capi.default_params(capi.SIGMA_DELTA)
        p.sd_amp_factor, p.sd_min_var, p.sd_max_var = amp, vmin, vmax
        o = pyoracle.Oracle(capi.SIGMA_DELTA, params=p)
        assert o.process(golden_frames[0]) == (None, None)  # SigmaDeltaBGS.cpp:33-39
        for t in range(1, len(golden_frames)):
            fg, bg = o.process(golden_frames[t])
            assert bg is None and np.array_equal(fg, g[tag][t - 1]), (tag, t)


@pytest.mark.skipif(not pyoracle.ref_sdlama_available(), reason="oracle/_ref is built only where /root/reference exists")
@pytest.mark.parametrize("amp,vmin,vmax", [(1, 15, 255), (2, 2, 255), (4, 10, 100), (1, 0, 255), (7, 3, 40), (2, 20, 300)])
def test_sigmadelta_oracle_matches_live_reference_build(amp, vmin, vmax):
    """Restatement vs the reference's own C file on adversarial clips: big jumps (int8 wrap of Mt - I), amplified
    differences above 255 (uint8 wrap of ++Vt), maxVar > 255 (truncated to uint8 by sdLaMa091's min/max)."""
    rng = np.random.default_rng(amp * 100 + vmin)
    frames = rng.integers(0, 256, (30, 40, 52, 3), dtype=np.uint8)
    frames[10:20] = (frames[10:20].astype(np.int32) // 8 + 200).astype(np.uint8)  # a long bright stretch
    want_all = pyoracle.ref_sigmadelta_clip(frames, amp, vmin, vmax)
    p = capi.default_params(capi.SIGMA_DELTA)
    p.sd_amp_factor, p.sd_min_var, p.sd_max_var = amp, vmin, vmax
    o = pyoracle.Oracle(capi.SIGMA_DELTA, params=p)
    for t, f in enumerate(frames):
        fg, _ = o.process(f)
        assert (t == 0) == (fg is None)
        if t:
            assert np.array_equal(fg, want_all[t - 1]), t


def test_lbsp_lut_values():
    lut3, lut1 = pyoracle.lbsp_lut(0.333, 0, 3), pyoracle.lbsp_lut(0.333, 0, 1)
    assert lut3[0] == 0 and lut3[255] == 85 and lut3[100] == 33       # saturate_cast<uchar>(t*0.333f), half-to-even
    assert lut1[255] == 28 and lut1[9] == 1                            # (t*0.333f)/3


def test_framediff_matches_independent_formula(golden_frames):
    want = np.load(os.path.join(GOLDEN, "framediff_indep.npz"))["masks"]
    o = pyoracle.Oracle(capi.FRAME_DIFF)
    assert o.process(golden_frames[0]) == (None, None)  # first frame: stored, outputs untouched
    for t in range(1, len(golden_frames)):
        fg, bg = o.process(golden_frames[t])
        assert bg is None
        assert np.array_equal(fg, want[t - 1])


@pytest.mark.parametrize("name,algo", [("fd", capi.FRAME_DIFF), ("sfd", capi.STATIC_FRAME_DIFF), ("wmm", capi.WMM), ("wmv", capi.WMV),
                                       ("abl", capi.ABL), ("asbl", capi.ASBL), ("mog2", capi.MOG2), ("mog1", capi.MOG1), ("sd", capi.SIGMA_DELTA), ("gmg", capi.GMG), ("subsense", capi.SUBSENSE),
                                       ("dpziv", capi.DP_ZIVKOVIC_AGMM), ("dpgrim", capi.DP_GRIMSON_GMM), ("dpwren", capi.DP_WREN_GA), ("dpmean", capi.DP_MEAN),
                                       ("dpmedian", capi.DP_ADAPTIVE_MEDIAN), ("lobster", capi.LOBSTER)])
def test_oracle_regression_vectors(name, algo, golden_frames, oracle_regress):
    o = pyoracle.Oracle(algo)
    want = oracle_regress[name + "_fg"]
    bg = None
    for t, f in enumerate(golden_frames):
        fg, b = o.process(f)
        if fg is None:
            assert (want[t] == UNTOUCHED).all()
        else:
            assert np.array_equal(fg, want[t]), (name, t)
        bg = b if b is not None else bg
    if name + "_bg_last" in oracle_regress.files:
        assert np.array_equal(bg, oracle_regress[name + "_bg_last"])
    if name == "mog2":
        n = golden_frames.shape[1] * golden_frames.shape[2]
        assert np.array_equal(o.get_state("w", (5, n), np.float32), oracle_regress["mog2_w"])
        assert np.array_equal(o.get_state("var", (5, n), np.float32), oracle_regress["mog2_var"])
        assert np.array_equal(o.get_state("mu", (5, 3, n), np.float32), oracle_regress["mog2_mu"])
        assert np.array_equal(o.get_state("nmodes", (n,), np.uint8), oracle_regress["mog2_nmodes"])


def test_gray_constants():
    """cv::cvtColor(BGR2GRAY): (B*1868 + G*9617 + R*4899 + 8192) >> 14; weights sum to 2^14 so gray(v,v,v) == v."""
    v = np.arange(256, dtype=np.uint8)
    img = np.stack([v, v, v], -1)[None]
    assert np.array_equal(pyoracle.bgr2gray(img)[0], v)
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255]]], np.uint8)
    assert pyoracle.bgr2gray(px)[0].tolist() == [29, 150, 76]


def test_mog2_first_frame_is_all_foreground(golden_frames):
    """On frame 1 every pixel spawns its first mode: not background, a == 1 -> shadow(127) -> thresholded to 255."""
    o = pyoracle.Oracle(capi.MOG2)
    fg, bg = o.process(golden_frames[0])
    assert (fg == 255).all()
    assert np.array_equal(bg, golden_frames[0])  # single mode of weight 1 centred on the pixel
    p = capi.default_params(capi.MOG2)
    p.enable_threshold = 0
    o = pyoracle.Oracle(capi.MOG2, params=p)
    fg, _ = o.process(golden_frames[0])
    black = (golden_frames[0] == 0).all(-1)  # detectShadowGMM: "no division by zero allowed" -> not a shadow -> 255
    assert (fg[~black] == 127).all() and (fg[black] == 255).all()


def test_mog2_invariants_on_saturating_clip():
    frames = synth.numpy_frames("sat", 30, 24, 40, seed=1234)
    o = pyoracle.Oracle(capi.MOG2)
    for f in frames:
        o.process(f, want_bg=False)
    n = 24 * 40
    w = o.get_state("w", (5, n), np.float32)
    var = o.get_state("var", (5, n), np.float32)
    nm = o.get_state("nmodes", (n,), np.uint8)
    assert nm.min() == 5, "S_sat must keep all 5 modes alive (it defines the dense 206 B/px traffic model)"
    assert (np.diff(w, axis=0) <= 0).all()
    assert (var >= 4).all() and (var <= 75).all()


def test_mog2_threads_do_not_change_results(golden_frames):
    a, b = pyoracle.Oracle(capi.MOG2, threads=1), pyoracle.Oracle(capi.MOG2, threads=4)
    for f in golden_frames[:8]:
        fa, ba = a.process(f)
        fb, bb = b.process(f)
        assert np.array_equal(fa, fb) and np.array_equal(ba, bb)


def test_mog1_first_frame_is_all_background(golden_frames):
    o = pyoracle.Oracle(capi.MOG1)
    fg, bg = o.process(golden_frames[0])
    assert (fg == 0).all() and bg is None  # BackgroundSubtractorMOG has no getBackgroundImage


def test_wrapper_quirks(golden_frames):
    """warm-up outputs untouched; FD/WMV never write a background; ASBL thresholds at 25 and is single-channel."""
    f = golden_frames
    o = pyoracle.Oracle(capi.WMV)
    assert o.process(f[0]) == (None, None) and o.process(f[1]) == (None, None)
    fg, bg = o.process(f[2])
    assert fg is not None and bg is None
    o = pyoracle.Oracle(capi.WMM)
    assert o.process(f[0]) == (None, None) and o.process(f[1]) == (None, None)
    fg, bg = o.process(f[2])
    assert fg is not None and bg.shape == f[0].shape
    o = pyoracle.Oracle(capi.ASBL)
    assert o.params.threshold == 25 and o.params.learning_frames == 90
    fg, bg = o.process(f[0])
    assert (fg == 0).all() and bg.ndim == 2 and np.array_equal(bg, pyoracle.bgr2gray(f[0]))
    o = pyoracle.Oracle(capi.STATIC_FRAME_DIFF)
    fg, bg = o.process(f[0])
    assert (fg == 0).all() and np.array_equal(bg, f[0])
    for t in range(1, 5):
        _, bg = o.process(f[t])
        assert np.array_equal(bg, f[0])  # frozen first frame


def test_abl_state_is_requantised_uint8(golden_frames):
    """AdaptiveBackgroundLearning keeps its background as uint8 (App. C 4): a constant scene is a fixed point."""
    o = pyoracle.Oracle(capi.ABL)
    for _ in range(5):
        fg, bg = o.process(golden_frames[0])
        assert np.array_equal(bg, golden_frames[0]) and (fg == 0).all()


def test_empty_input_and_geometry_change():
    o = pyoracle.Oracle(capi.FRAME_DIFF)
    assert o.process(None) == (None, None)
    a = synth.random_frames(2, 8, 8, 3, seed=1)
    o.process(a[0])
    with pytest.raises(RuntimeError):
        o.process(a[1][:4])


def test_morphology_primitives():
    m = np.zeros((9, 9), np.uint8)
    m[3:6, 3:6] = 255
    assert pyoracle.erode3x3(m)[4, 4] == 255 and pyoracle.erode3x3(m).sum() == 255
    assert pyoracle.dilate3x3(m)[2:7, 2:7].min() == 255 and pyoracle.dilate3x3(m).sum() == 25 * 255
    ones = np.full((4, 4), 255, np.uint8)
    assert (pyoracle.erode3x3(ones, 3) == 255).all()  # outside pixels do not take part in the erosion
    hole = np.full((7, 7), 255, np.uint8)
    hole[0, :] = 0
    hole[3, 3] = 0
    ff = pyoracle.floodfill_from_origin(hole, 255)
    assert ff[0, 0] == 255 and ff[3, 3] == 0  # the enclosed hole is not reached from (0,0)
    rng = np.random.default_rng(0)
    g = rng.integers(0, 256, (12, 15), dtype=np.uint8)
    p = np.pad(g, 1, mode="edge")
    want = np.array([[np.median(p[y:y + 3, x:x + 3]) for x in range(15)] for y in range(12)]).astype(np.uint8)
    assert np.array_equal(pyoracle.median_blur(g, 3), want)


def test_subsense_oracle_basics(golden_frames):
    """SuBSENSE restatement: the first call initialises AND classifies (SuBSENSE.cpp:27-38), the 2-px border never fires,
    the background image is the mean of the 50 colour samples, and the counter-based RNG makes runs reproducible."""
    a, b = pyoracle.Oracle(capi.SUBSENSE), pyoracle.Oracle(capi.SUBSENSE)
    for t, f in enumerate(golden_frames[:8]):
        fa, ba = a.process(f)
        fb, bb = b.process(f)
        assert fa is not None and ba is not None
        assert np.array_equal(fa, fb) and np.array_equal(ba, bb)
        assert set(np.unique(fa)) <= {0, 255}
    n = golden_frames.shape[1] * golden_frames.shape[2]
    color = a.get_state("color", (50, n, 3), np.uint8).astype(np.float32)
    acc = np.zeros((n, 3), np.float32)
    for k in range(50):
        acc += color[k] / np.float32(50)
    assert np.array_equal(ba.reshape(n, 3), np.clip(np.rint(acc), 0, 255).astype(np.uint8))
    R = a.get_state("R", (n,), np.float32)
    assert R.min() >= 1.0
    sc = a.get_state("scalars", (7,), np.float64)
    assert sc[0] == 8 and sc[3] == 4.0 and sc[4] == 512.0  # below QVGA: caps doubled, no learning-rate scaling


def test_subsense_static_scene_goes_quiet():
    rng = np.random.default_rng(3)
    base = rng.integers(40, 200, (40, 56, 3)).astype(np.int32)
    o = pyoracle.Oracle(capi.SUBSENSE)
    for t in range(15):
        f = np.clip(base + rng.integers(-2, 3, base.shape), 0, 255).astype(np.uint8)
        fg, _ = o.process(f)
    assert (fg == 255).mean() < 0.01


def test_counter_rng_is_shared_by_both_sides():
    """ss_rand() is the whole stochastic contract: same function in oracle/subsense_oracle.c and kernel_subsense.h."""
    import ctypes as C
    l = pyoracle.lib()
    l.ss_rand.restype = C.c_uint32
    l.ss_rand.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]

    def ref(frame, pixel, draw):
        M = 0xFFFFFFFF
        x = (frame * 0x9E3779B1) & M
        x ^= (pixel + 0x85EBCA6B + ((x << 6) & M) + (x >> 2)) & M
        x ^= ((draw + 1) * 0xC2B2AE35) & M
        x ^= x >> 16
        x = (x * 0x85EBCA6B) & M
        x ^= x >> 13
        x = (x * 0xC2B2AE35) & M
        x ^= x >> 16
        return x >> 1

    for args in [(0, 0, 0), (1, 2, 3), (77, 123456, 6), (4000000000, 0xFFFFFFFF, 16)]:
        assert l.ss_rand(*args) == ref(*args)
    vals = np.array([l.ss_rand(5, p, 4) % 8 for p in range(8000)])
    assert np.bincount(vals, minlength=8).min() > 800  # roughly uniform


def test_subsense_oracle_grayscale(golden_gray):
    o = pyoracle.Oracle(capi.SUBSENSE)
    for f in golden_gray:
        fg, bg = o.process(f)
        assert fg.shape == f.shape and bg.shape == f.shape
    lut = o.get_state("lut", (256,), np.uint8)
    assert lut[255] <= 28  # (t * 0.333f) / 3, possibly auto-decremented since


def test_gmg_oracle_phases(golden_frames):
    """GMG: 20 training frames without any foreground (GMG.cpp:44), then Bayesian decisions; no background image ever."""
    o = pyoracle.Oracle(capi.GMG)
    frames = np.concatenate([golden_frames, golden_frames[::-1]])
    for t, f in enumerate(frames):
        fg, bg = o.process(f)
        assert bg is None
        if t < 20:
            assert (fg == 0).all()
    assert 0.0 < (fg == 255).mean() < 0.6
    n = f.shape[0] * f.shape[1]
    nf = o.get_state("nfeatures", (n,), np.int32)
    w = o.get_state("weights", (64, n), np.float32)
    assert nf.min() >= 1 and nf.max() <= 64
    assert np.allclose(w.sum(0), 1.0, atol=0.05)  # histograms stay (nearly) normalised


def test_unit_absdiff_is_integer_absdiff():
    """The reference converts frames to float/255, takes |I - B| and converts back (AdaptiveBackgroundLearning.cpp:50,64;
    AdaptiveSelectiveBackgroundLearning.cpp:50-57).  For all 65 536 byte pairs that round trip equals |i - b|: the kernels use
    the integer form (abl_kernel, asbl_kernel); this is the exhaustive check that licenses it."""
    sf = np.float32(1.0 / 255.0)
    i = np.arange(256, dtype=np.float32)[:, None] * sf
    b = np.arange(256, dtype=np.float32)[None, :] * sf
    v = (np.abs(i - b).astype(np.float32) * np.float32(255.0)).astype(np.float32)
    got = np.clip(np.rint(v), 0, 255).astype(np.int64)  # cv::saturate_cast<uchar>(float): cvRound, ties to even
    want = np.abs(np.arange(256)[:, None] - np.arange(256)[None, :])
    assert np.array_equal(got, want)
    assert float(np.min(np.abs(v - np.floor(v) - 0.5))) > 0.4  # nowhere near a rounding tie


@pytest.mark.parametrize("connectivity", [8, 4])
def test_components_oracle_vs_scipy(connectivity):
    """The oracle's connected components against an independent implementation (scipy.ndimage.label + find_objects):
    same partition, same bounding boxes and areas, roots = first pixel in raster order, boxes sorted by root."""
    from scipy import ndimage
    rng = np.random.default_rng(connectivity)
    structure = np.ones((3, 3), int) if connectivity == 8 else None
    for shape, density in (((40, 60), 0.3), ((40, 60), 0.55), ((1, 50), 0.5), ((33, 1), 0.5), ((64, 64), 0.0), ((64, 64), 1.0)):
        m = np.where(rng.random(shape) < density, 255, 0).astype(np.uint8)
        labels, boxes, n = pyoracle.components(m, connectivity)
        ref, nref = ndimage.label(m, structure=structure)
        assert n == nref == len(boxes)
        assert np.array_equal(labels >= 0, m != 0)
        # same partition: scipy label -> our root must be a bijection, and the root is the component's smallest raster index
        for k, sl in enumerate(ndimage.find_objects(ref), start=1):
            sel = ref == k
            roots = np.unique(labels[sel])
            assert len(roots) == 1 and roots[0] == np.flatnonzero(sel.ravel())[0]
            b = boxes[boxes["root"] == roots[0]][0]
            assert (b["y"], b["x"], b["h"], b["w"]) == (sl[0].start, sl[1].start, sl[0].stop - sl[0].start, sl[1].stop - sl[1].start)
            assert b["area"] == int(sel.sum())
        assert np.all(np.diff(boxes["root"]) > 0)


def test_dp_oracle_invariants(golden_frames):
    """package_bgs/dp restatement (dp_oracle.c): properties the sources guarantee, checked on the golden clip.
    Zivkovic: weights of the used modes sum to 1 and are sorted descending (ZivkovicAGMM.cpp:219-234, 259-263), variances stay in
    [4, 180] (:192); first frame: every pixel creates its first mode, and an empty model has no background gaussian -> all
    foreground.  Grimson: modes sorted by weight/sqrt(variance).  WrenGA: variance clamp.  AdaptiveMedian: the median moves
    by at most one grey level and only on frames with frame % samplingRate == 1 (AdaptiveMedianBGS.cpp:60)."""
    n = golden_frames.shape[1] * golden_frames.shape[2]
    for algo, F in ((capi.DP_ZIVKOVIC_AGMM, 5), (capi.DP_GRIMSON_GMM, 6)):
        o = pyoracle.Oracle(algo)
        fg, bg = o.process(golden_frames[0])
        assert bg is None and (fg == 255).all()
        assert (o.get_state("nmodes", (n,), np.uint8) == 1).all()
        for f in golden_frames[1:]:
            o.process(f)
        m = o.get_state("modes", (3, F, n), np.float32)
        nm = o.get_state("nmodes", (n,), np.uint8)
        assert nm.min() >= 1 and nm.max() <= 3
        used = np.arange(3)[:, None] < nm[None, :]
        w = np.where(used, m[:, 4], 0)
        assert np.allclose(w.sum(0), 1, atol=1e-5)
        var = m[:, 0][used]
        assert var.min() >= 4 and var.max() <= 180
        key = m[:, 4] if F == 5 else m[:, 5]
        for k in range(2):
            both = used[k] & used[k + 1]
            assert (key[k][both] >= key[k + 1][both]).all()
    o = pyoracle.Oracle(capi.DP_WREN_GA)
    fg, _ = o.process(golden_frames[0])
    assert (fg == 0).all()  # the model IS the first frame
    for f in golden_frames[1:]:
        o.process(f)
    g = o.get_state("gauss", (4, n), np.float32)
    assert g[3].min() >= 4 and g[3].max() <= 180
    o = pyoracle.Oracle(capi.DP_ADAPTIVE_MEDIAN)
    prev = None
    for t, f in enumerate(golden_frames):
        o.process(f)
        med = o.get_state("median", (n * 3,), np.uint8).astype(int)
        if prev is not None:
            step = np.abs(med - prev).max()
            assert step <= 1 and (step == 0 or t % 7 == 1)
        prev = med


def test_morphology_oracle_vs_scipy():
    """The oracle's cv::-style primitives against independent implementations: median (BORDER_REPLICATE) = scipy median_filter
    mode='nearest'; erode/dilate with cells outside the image ignored = grey_erosion/dilation with cval 255/0; n iterations of
    3x3 = one (2n+1)^2 box; floodFill from the origin = the 4-connected region of equal value containing (0,0)."""
    from scipy import ndimage
    rng = np.random.default_rng(3)
    for shape in ((40, 57), (9, 9), (1, 30), (25, 2)):
        g = rng.integers(0, 256, shape, dtype=np.uint8)
        b = np.where(rng.random(shape) < 0.45, 255, 0).astype(np.uint8)
        for k in (3, 5, 9):
            assert np.array_equal(pyoracle.median_blur(g, k), ndimage.median_filter(g, size=k, mode="nearest")), (shape, k)
            assert np.array_equal(pyoracle.median_blur(b, k), ndimage.median_filter(b, size=k, mode="nearest")), (shape, k)
        for it in (1, 2, 3):
            sz = 2 * it + 1
            assert np.array_equal(pyoracle.erode3x3(g, it), ndimage.grey_erosion(g, size=(sz, sz), mode="constant", cval=255)), (shape, it)
            assert np.array_equal(pyoracle.dilate3x3(g, it), ndimage.grey_dilation(g, size=(sz, sz), mode="constant", cval=0)), (shape, it)
        lab, _ = ndimage.label(b == b[0, 0])  # 4-connected by default
        want = b.copy()
        want[lab == lab[0, 0]] = 255
        assert np.array_equal(pyoracle.floodfill_from_origin(b, 255), want), shape


# ---- N3 frame preparation (oracle/ingest_oracle.c): the restated OpenCV pieces against independent formulations ----

def _ingest_cfg(**kw):
    from tracking_amd import capi
    return capi.default_ingest(**kw)


def test_ingest_gaussian_kernel_takes_the_fixed_point_path():
    """cv::GaussianBlur on 8U only runs in fixed point when getKernelType calls the float kernel smooth (|sum - 1| within FLT_EPSILON):
    the restatement checks that condition for (7, 1.5) and this pins the integer kernel it derives."""
    ik, smooth = pyoracle.gaussian7_kernel()
    assert smooth
    assert ik == [9, 28, 55, 69, 55, 28, 9]  # sums to 253, not 256: each 8-bit coefficient is rounded on its own (as recalled, nothing renormalises)
    want = np.exp(-0.5 * (np.arange(7) - 3.0) ** 2 / 1.5 ** 2)
    assert ik == [int(round(v)) for v in want / want.sum() * 256]


@pytest.mark.parametrize("shape", [(40, 60, 3), (9, 5, 1), (1, 1, 3), (3, 70, 3), (64, 64, 1)])
def test_ingest_blur_vs_scipy(shape):
    from scipy import ndimage
    rng = np.random.default_rng(shape[0] * 100 + shape[1])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    got = pyoracle.ingest(_ingest_cfg(gaussian_blur=1), img if shape[2] == 3 else img[:, :, 0])
    ik, _ = pyoracle.gaussian7_kernel()
    t = ndimage.correlate1d(img.astype(np.int64), np.array(ik, np.int64), axis=1, mode="mirror")   # BORDER_REFLECT_101
    t = ndimage.correlate1d(t, np.array(ik, np.int64), axis=0, mode="mirror")
    want = np.clip((t + (1 << 15)) >> 16, 0, 255).astype(np.uint8)
    assert np.array_equal(got.reshape(shape), want)


def test_ingest_equalize_vs_numpy():
    rng = np.random.default_rng(3)
    for img in (rng.integers(30, 200, (50, 70), dtype=np.uint8), np.full((8, 9), 77, np.uint8), (rng.random((40, 40)) < 0.5).astype(np.uint8) * 255):
        got = pyoracle.ingest(_ingest_cfg(equalize_hist=1), img)
        hist = np.bincount(img.ravel(), minlength=256)
        i = int(np.flatnonzero(hist)[0])
        if hist[i] == img.size:
            want = np.full_like(img, i)
        else:
            scale = np.float32(255.0) / np.float32(img.size - hist[i])
            csum = np.cumsum(hist[i + 1:]).astype(np.float32)
            lut = np.zeros(256, np.uint8)
            lut[i + 1:] = np.clip(np.rint(csum * scale), 0, 255).astype(np.uint8)
            want = lut[img]
        assert np.array_equal(got, want)
    assert pyoracle.ingest(_ingest_cfg(equalize_hist=1), np.zeros((4, 4, 3), np.uint8)) is None  # cv::equalizeHist asserts CV_8UC1


def test_ingest_flip_roi_and_identity_resize_are_exact():
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(pyoracle.ingest(_ingest_cfg(), img), img)  # cvResize to the same size: every coefficient is (2048, 0)
    assert np.array_equal(pyoracle.ingest(_ingest_cfg(flip=1), img), img[::-1])
    c = _ingest_cfg(flip=1, roi_x0=5, roi_y0=7, roi_x1=40, roi_y1=30)
    assert np.array_equal(pyoracle.ingest(c, img), img[::-1][7:30, 5:40])
    assert pyoracle.ingest(_ingest_cfg(roi_x0=5, roi_y0=7, roi_x1=60, roi_y1=30), img) is None  # cvSetImageROI outside the frame


@pytest.mark.parametrize("pct", [50, 25, 75, 130, 33])
def test_ingest_resize_vs_float_bilinear(pct):
    """R1 against the textbook float formula (half-pixel centres, clamped taps): the 11-bit fixed point stays within 1 grey level;
    50 % of an even-sized frame is exactly the rounded 2x2 block mean (INTER_AREA's fast path)."""
    rng = np.random.default_rng(pct)
    img = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    got = pyoracle.ingest(_ingest_cfg(resize_percent=pct), img)
    rh, rw = 48 * pct // 100, 64 * pct // 100
    assert got.shape == (rh, rw, 3)
    if pct == 50:
        b = img.astype(np.int32)
        assert np.array_equal(got, ((b[0::2, 0::2] + b[0::2, 1::2] + b[1::2, 0::2] + b[1::2, 1::2] + 2) >> 2).astype(np.uint8))
        return
    ys = (np.arange(rh) + 0.5) * (48 / rh) - 0.5
    xs = (np.arange(rw) + 0.5) * (64 / rw) - 0.5
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    yc0, yc1 = np.clip(y0, 0, 47), np.clip(y0 + 1, 0, 47)
    xc0, xc1 = np.clip(x0, 0, 63), np.clip(x0 + 1, 0, 63)
    f = img.astype(np.float64)
    want = (f[yc0][:, xc0] * (1 - fx) + f[yc0][:, xc1] * fx) * (1 - fy) + (f[yc1][:, xc0] * (1 - fx) + f[yc1][:, xc1] * fx) * fy
    assert np.max(np.abs(got.astype(np.float64) - want)) <= 1.0


def test_cpu_code_is_clean_under_asan_and_ubsan():
    """tools/sanitize_cpu.sh: the oracle's C code (through these same CPU tests) and the host-side C++ mirror (XML handling, set-up, the
    no-GPU exception path) under AddressSanitizer + UndefinedBehaviorSanitizer.  GPU sanitizers do not exist on this pool; this is the
    CPU half."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not shutil.which("gcc") or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan in this image")
    if os.environ.get("BGS_ORACLE_LIB"):
        pytest.skip("already running inside the sanitizer run")
    r = subprocess.run(["bash", os.path.join(root, "tools", "sanitize_cpu.sh")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "oracle under ASan+UBSan: OK" in r.stdout and "host mirror under ASan+UBSan: OK" in r.stdout


@pytest.mark.parametrize("shape", [(243, 325), (480, 854), (250, 333), (241, 320)])
def test_resize_area_general_path_vs_exact_area_average(shape):
    """cv::resize(INTER_AREA) to (w/8, h/8) for sizes that are not multiples of 8 (SuBSENSE's frame-level block,
    BackgroundSubtractorSuBSENSE.cpp:153, :656): the restated float accumulation (recalled from OpenCV 2.4 resizeArea_, unpinned) must
    stay within one grey level of the exact area-weighted mean computed in float64 (the weights of a destination cell sum to 1; the
    last cell of an axis is the clipped one)."""
    rng = np.random.default_rng(shape[0])
    H, W = shape
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    dh, dw = H // 8, W // 8
    got = pyoracle.resize_area(img, dh, dw).astype(np.float64)

    def weights(ssize, dsize):
        scale = ssize / dsize
        M = np.zeros((dsize, ssize))
        for d in range(dsize):
            a, b = d * scale, min((d + 1) * scale, ssize)
            for sx in range(int(np.floor(a)), min(int(np.ceil(b)), ssize)):
                M[d, sx] = max(0.0, min(b, sx + 1) - max(a, sx))
            M[d] /= M[d].sum()
        return M
    My, Mx = weights(H, dh), weights(W, dw)
    exact = np.einsum("yh,hwc,xw->yxc", My, img.astype(np.float64), Mx)
    assert np.abs(got - exact).max() <= 1.0, float(np.abs(got - exact).max())
    assert np.abs(got - np.rint(exact)).mean() < 0.02  # nearly always THE rounded mean
    g1 = pyoracle.resize_area(img[:, :, 1].copy(), dh, dw)
    assert np.array_equal(g1, got[:, :, 1].astype(np.uint8))  # the 1-channel path is the same arithmetic
