"""GPU parity: libbgs_hip (through the C ABI) vs the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): uint8 masks / backgrounds bit-exact, float model state within 1e-4.
All tests here need a real MI355X: `pytest -m gpu`.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import pyoracle
from tools import synth
from tracking_amd import Engine, capi

from gpu_helpers import *  # noqa: F401,F403
from gpu_helpers import _params, _torch, _cc_masks


pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(ALGOS))
def test_golden_frames(name, golden_frames):
    """The reference's own frames/*.png crop, 24 consecutive frames."""
    eng, orc, _ = run_pair(ALGOS[name], golden_frames)
    check_state(name, eng, orc, golden_frames.shape[1] * golden_frames.shape[2])


@pytest.mark.parametrize("name", sorted(ALGOS))
@pytest.mark.parametrize("shape", [(48, 64), (37, 53), (5, 7), (1, 1), (64, 256)])
def test_seeded_random(name, shape):
    """vector (16 px/lane), dword (4 px/lane) and ragged (1 px/lane) kernels all hit: 64x256, 48x64 | 37x53 ..."""
    frames = synth.random_frames(10, shape[0], shape[1], 3, seed=hash((name, shape)) % 1000)
    eng, orc, _ = run_pair(ALGOS[name], frames)
    check_state(name, eng, orc, shape[0] * shape[1])


@pytest.mark.parametrize("name", ["FrameDifferenceBGS", "StaticFrameDifferenceBGS", "WeightedMovingMeanBGS", "WeightedMovingVarianceBGS", "AdaptiveBackgroundLearning",
                                  "AdaptiveSelectiveBackgroundLearning", "MixtureOfGaussianV1BGS"])
def test_single_channel(name, golden_gray):
    eng, orc, _ = run_pair(ALGOS[name], golden_gray)
    if name == "MixtureOfGaussianV1BGS":
        check_mog1_state(eng, orc, golden_gray.shape[1] * golden_gray.shape[2], C=1)


def test_sigmadelta_matches_reference_fixture(golden_frames):
    """GPU vs masks produced by the reference's own sdLaMa091.cpp (tests/golden/sigmadelta_ref.npz): pinned parity."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sigmadelta_ref.npz"))
    for tag, (amp, vmin, vmax) in {"default": (1, 15, 255), "amp3": (3, 2, 200)}.items():
        eng = Engine(capi.SIGMA_DELTA, params=_params(capi.SIGMA_DELTA, sd_amp_factor=amp, sd_min_var=vmin, sd_max_var=vmax))
        assert eng.process(golden_frames[0]) == (None, None)
        for t in range(1, len(golden_frames)):
            fg, bg = eng.process(golden_frames[t])
            assert bg is None and np.array_equal(fg, g[tag][t - 1]), (tag, t)


@pytest.mark.parametrize("kw", [dict(sd_amp_factor=2, sd_min_var=2), dict(sd_amp_factor=7, sd_min_var=3, sd_max_var=40), dict(sd_max_var=300), dict(sd_min_var=0)])
def test_sigmadelta_variants(kw):
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, (20, 33, 47, 3), dtype=np.uint8)
    eng, orc, _ = run_pair(capi.SIGMA_DELTA, frames, params=_params(capi.SIGMA_DELTA, **kw))
    check_state("SigmaDeltaBGS", eng, orc, 33 * 47)


def test_sigmadelta_rejects_gray(golden_gray):
    with pytest.raises(capi.BgsError) as ei:
        Engine(capi.SIGMA_DELTA).process(golden_gray[0])
    assert ei.value.code == capi.ERR_UNSUPPORTED


def test_mog2_rejects_gray(golden_gray):
    """cv::BackgroundSubtractorMOG2::getBackgroundImage asserts nchannels == 3; the wrapper calls it every frame."""
    eng = Engine(capi.MOG2)
    with pytest.raises(capi.BgsError) as ei:
        eng.process(golden_gray[0])
    assert ei.value.code == capi.ERR_UNSUPPORTED


@pytest.mark.parametrize("name", sorted(ALGOS))
def test_empty_input_is_a_noop(name):
    """`if(img_input.empty()) return;` — outputs untouched, no state change."""
    eng = Engine(ALGOS[name])
    assert eng.process(None) == (None, None)
    assert eng.frames_seen() == 0
    frames = synth.random_frames(4, 16, 16, 3, seed=3)
    eng.process(frames[0])
    assert eng.process(np.empty((0, 0, 3), np.uint8)) == (None, None)
    assert eng.frames_seen() == 1


def test_warmup_outputs_untouched():
    """FD frame 1, WMM/WMV frames 1-2 return with outputs untouched; FD/WMV never write a background."""
    frames = synth.random_frames(4, 16, 32, 3, seed=5)
    for algo, warm, has_bg in ((capi.FRAME_DIFF, 1, False), (capi.WMM, 2, True), (capi.WMV, 2, False), (capi.STATIC_FRAME_DIFF, 0, True), (capi.ABL, 0, True),
                               (capi.ASBL, 0, True), (capi.MOG1, 0, False), (capi.MOG2, 0, True), (capi.SIGMA_DELTA, 1, False), (capi.GMG, 0, False),
                               (capi.DP_ZIVKOVIC_AGMM, 0, False), (capi.DP_GRIMSON_GMM, 0, False), (capi.DP_WREN_GA, 0, False), (capi.DP_MEAN, 0, False),
                               (capi.DP_ADAPTIVE_MEDIAN, 0, False)):
        eng = Engine(algo)
        for t, f in enumerate(frames):
            fg, bg = eng.process(f)
            assert (fg is None) == (t < warm), (algo, t)
            assert (bg is None) == (t < warm or not has_bg), (algo, t)


def test_strided_roi_input(golden_frames):
    """VideoCapture hands FrameProcessor a ROI view whose row step exceeds 3*cols (VideoCapture.cpp:203-209)."""
    for algo in (capi.FRAME_DIFF, capi.MOG2, capi.ABL):
        eng, orc = Engine(algo), pyoracle.Oracle(algo)
        for f in golden_frames[:6]:
            roi = f[10:50, 20:75]  # non-contiguous view
            assert not roi.flags["C_CONTIGUOUS"]
            fg, bg = eng.process(roi)
            ofg, obg = orc.process(np.ascontiguousarray(roi))
            if ofg is not None:
                assert np.array_equal(fg, ofg)
            if obg is not None:
                assert np.array_equal(bg, obg)


def test_geometry_change_is_an_error(golden_frames):
    eng = Engine(capi.FRAME_DIFF)
    eng.process(golden_frames[0])
    with pytest.raises(capi.BgsError) as ei:
        eng.process(golden_frames[1][:40])
    assert ei.value.code == capi.ERR_GEOMETRY


@pytest.mark.parametrize("kw", [dict(enable_threshold=0), dict(threshold=40), dict(threshold=255), dict(threshold=-1)])
@pytest.mark.parametrize("name", sorted(ALGOS))
def test_wrapper_threshold_variants(name, kw, golden_frames):
    run_pair(ALGOS[name], golden_frames[:8], params=_params(ALGOS[name], **kw))


def test_unweighted_variants(golden_frames):
    """enableWeight=0: WMM uses (A+B+C)/3.0, WMV uses 0.3/0.3/0.3 (sic)."""
    for algo in (capi.WMM, capi.WMV):
        run_pair(algo, golden_frames[:8], params=_params(algo, enable_weight=0))


@pytest.mark.parametrize("kw", [dict(alpha=0.2), dict(alpha=0.0), dict(alpha=1.0), dict(limit=3), dict(limit=0)])
def test_abl_variants(kw, golden_frames):
    run_pair(capi.ABL, golden_frames[:10], params=_params(capi.ABL, **kw))


@pytest.mark.parametrize("kw", [dict(alpha=0.005), dict(alpha=0.3), dict(alpha=-1.0), dict(alpha=1.0),
                                dict(enable_threshold=0), dict(enable_threshold=0, mog2_detect_shadows=0),
                                dict(threshold=200), dict(mog2_var_threshold=4.0, mog2_var_threshold_gen=2.0),
                                dict(mog2_ct=0.6), dict(mog2_background_ratio=0.5), dict(mog2_var_init=50.0, mog2_var_max=60.0)])
def test_mog2_param_variants(kw, golden_frames):
    """alpha<0 = OpenCV's automatic 1/min(2n,history) rate; alpha>=1 re-initialises every frame; threshold=200 sits between
    the shadow value 127 and 255 so shadow detection changes the delivered mask; ct=0.6 makes matched modes prunable."""
    p = _params(capi.MOG2, **kw)
    eng, orc, _ = run_pair(capi.MOG2, golden_frames[:12], params=p)
    check_mog2_state(eng, orc, golden_frames.shape[1] * golden_frames.shape[2])


def test_mog2_saturating_clip_uses_all_modes():
    """S_sat (the roofline workload): 5 live modes per pixel, constant re-sorting, new-mode replacement."""
    frames = synth.numpy_frames("sat", 40, 32, 64, seed=1234)
    eng, orc, _ = run_pair(capi.MOG2, frames)
    n = 32 * 64
    check_mog2_state(eng, orc, n)
    assert eng.get_state("nmodes", (n,), np.uint8).min() == 5


def test_mog2_long_run_static_scene():
    """surveillance-like clip long enough for modes to be pruned and re-created."""
    frames = synth.numpy_frames("surv", 120, 48, 80, seed=4321)
    eng, orc, _ = run_pair(capi.MOG2, frames, want_bg=True)
    check_mog2_state(eng, orc, 48 * 80)


def test_params_can_change_between_frames(golden_frames):
    """loadConfig() runs at the top of every process(): thresholds / alpha may change mid-stream."""
    eng, orc = Engine(capi.MOG2), pyoracle.Oracle(capi.MOG2)
    for t, f in enumerate(golden_frames[:10]):
        if t == 4:
            p = _params(capi.MOG2, alpha=0.2, threshold=130)
            eng.set_params(p)
            orc.set_params(p)
        fg, bg = eng.process(f)
        ofg, obg = orc.process(f)
        assert np.array_equal(fg, ofg) and np.array_equal(bg, obg)


def test_streams_are_independent(golden_frames):
    """One engine, 3 streams fed different clips in interleaved order == 3 separate oracles."""
    clips = [golden_frames[0:8], golden_frames[8:16], golden_frames[16:24][::-1]]
    for algo in (capi.MOG2, capi.WMV, capi.ABL, capi.ASBL, capi.MOG1, capi.GMG):
        eng = Engine(algo, n_streams=3)
        orcs = [pyoracle.Oracle(algo) for _ in clips]
        for t in range(8):
            for s in (2, 0, 1):
                fg, bg = eng.process(clips[s][t], stream=s)
                ofg, obg = orcs[s].process(clips[s][t])
                assert (fg is None) == (ofg is None)
                if fg is not None:
                    assert np.array_equal(fg, ofg), (algo, t, s)
                if obg is not None:
                    assert np.array_equal(bg, obg), (algo, t, s)
        if algo == capi.MOG2:
            for s in range(3):
                check_mog2_state(eng, orcs[s], golden_frames.shape[1] * golden_frames.shape[2], stream=s)


# ----------------------------------------------------------------------------- device (roofline) path


@pytest.mark.parametrize("name", sorted(ALGOS))
@pytest.mark.parametrize("borrow", [False, True])
def test_device_batch_matches_oracle(name, borrow):
    """bgs_process_batch_device: S streams x pixels in one launch, byte mask + bit-packed mask + background."""
    torch = _torch()
    algo = ALGOS[name]
    S, T, H, W = 4, 9, 32, 64
    clips = np.stack([synth.random_frames(T, H, W, 3, seed=100 + s) for s in range(S)])  # [S][T][H][W][3]
    eng = Engine(algo, n_streams=S)
    eng.set_geometry(H, W, 3)
    if borrow:
        eng.set_option(capi.OPT_BORROW_FRAMES, 1)
    orcs = [pyoracle.Oracle(algo) for _ in range(S)]
    keep = []
    for t in range(T):
        d_frames = torch.from_numpy(np.ascontiguousarray(clips[:, t])).cuda()
        keep.append(d_frames)  # borrowed history must stay alive
        d_fg = torch.full((S, H, W), 9, dtype=torch.uint8, device="cuda")
        d_bg = torch.full((S, H, W, 1 if algo == capi.ASBL else 3), 9, dtype=torch.uint8, device="cuda")
        has_bits = True  # (the stencil / median / sample-consensus paths pack the finished byte mask)
        d_bits = torch.zeros((S, H * W // 64), dtype=torch.int64, device="cuda") if has_bits else None
        flags = eng.process_batch_device(d_frames, d_fg, d_bg, d_bits)
        torch.cuda.synchronize()
        fg, bg = d_fg.cpu().numpy(), d_bg.cpu().numpy()
        bits = np.unpackbits(d_bits.cpu().numpy().view(np.uint8).reshape(S, -1), axis=1, bitorder="little").reshape(S, H, W) if has_bits else None
        for s in range(S):
            ofg, obg = orcs[s].process(clips[s, t])
            assert bool(flags & capi.FG_VALID) == (ofg is not None)
            assert bool(flags & capi.BG_VALID) == (obg is not None)
            if ofg is not None:
                assert np.array_equal(fg[s], ofg), (t, s)
                if has_bits:
                    assert np.array_equal(bits[s] * 255, np.where(ofg != 0, 255, 0)), (t, s)
            else:
                assert (fg[s] == 9).all()  # untouched
            if obg is not None:
                assert np.array_equal(bg[s].reshape(obg.shape), obg), (t, s)
    if algo == capi.MOG2:
        for s in range(S):
            check_mog2_state(eng, orcs[s], H * W, stream=s)


RAGGED = ["MixtureOfGaussianV2BGS", "MixtureOfGaussianV1BGS", "FrameDifferenceBGS", "WeightedMovingVarianceBGS", "AdaptiveBackgroundLearning",
          "AdaptiveSelectiveBackgroundLearning", "GMG", "SigmaDeltaBGS", "DPZivkovicAGMMBGS", "DPAdaptiveMedianBGS", "SuBSENSEBGS", "LOBSTERBGS"]


@pytest.mark.parametrize("name", RAGGED)
@pytest.mark.parametrize("with_fg", [True, False])
def test_packed_masks_of_frames_that_are_not_a_multiple_of_64_pixels(name, with_fg):
    """350 x 233 = 81 550 pixels = 1 274 words + 14 bits (VideoCapture resizes by a percentage, VideoCapture.cpp:158-207: any size can
    reach IBGS::process).  Stream k of a call owns words [k W, (k + 1) W), W = ceil(n / 64); the bits of its last word past pixel
    n - 1 are zero.  Per-frame calls over 3 streams, then one clip call; with and without a byte mask beside the packed one."""
    torch = _torch()
    algo = dict(ALGOS, SuBSENSEBGS=capi.SUBSENSE, LOBSTERBGS=capi.LOBSTER)[name]
    S, T, NC, H, W = 3, 5, 3, 233, 350
    n, words = H * W, (H * W + 63) // 64
    clips = np.stack([synth.random_frames(T + NC, H, W, 3, seed=640 + s) for s in range(S)])
    eng = Engine(algo, n_streams=S)
    eng.set_geometry(H, W, 3)
    orcs = [pyoracle.Oracle(algo) for _ in range(S)]

    def unpack(t_bits):
        raw = t_bits.cpu().numpy().view(np.uint8).reshape(-1, words * 8)
        bits = np.unpackbits(raw, axis=1, bitorder="little")
        assert not bits[:, n:].any(), "tail bits of the last word must be zero"
        return bits[:, :n].reshape(-1, H, W)

    for t in range(T):
        d_frames = torch.from_numpy(np.ascontiguousarray(clips[:, t])).cuda()
        d_fg = torch.full((S, H, W), 9, dtype=torch.uint8, device="cuda") if with_fg else None
        d_bits = torch.full((S, words), -1, dtype=torch.int64, device="cuda")
        flags = eng.process_batch_device(d_frames, d_fg, None, d_bits)
        torch.cuda.synchronize()
        for s in range(S):
            ofg, _ = orcs[s].process(clips[s, t], want_bg=False)
            assert bool(flags & capi.FG_VALID) == (ofg is not None), (name, t)
            if ofg is None:
                assert bool((d_bits[s] == -1).all()), (name, t, "a warm-up frame leaves the words untouched")
                continue
            assert np.array_equal(unpack(d_bits[s:s + 1])[0] * 255, np.where(ofg != 0, 255, 0)), (name, t, s)
            if with_fg:
                assert np.array_equal(d_fg[s].cpu().numpy(), ofg), (name, t, s)
    slab = torch.from_numpy(np.ascontiguousarray(clips[:, T:T + NC].transpose(1, 0, 2, 3, 4))).cuda()  # [NC][S][H][W][3]
    d_fgc = torch.full((NC, S, H, W), 9, dtype=torch.uint8, device="cuda") if with_fg else None
    d_bitsc = torch.full((NC, S, words), -1, dtype=torch.int64, device="cuda")
    eng.process_clip_device(slab, NC, d_fgc, None, d_bitsc)
    torch.cuda.synchronize()
    got = unpack(d_bitsc.reshape(NC * S, words)).reshape(NC, S, H, W)
    for t in range(NC):
        for s in range(S):
            ofg, _ = orcs[s].process(clips[s, T + t], want_bg=False)
            assert np.array_equal(got[t, s] * 255, np.where(ofg != 0, 255, 0)), (name, "clip", t, s)
    eng.close()


@pytest.mark.parametrize("px", [1, 2, 4])
def test_mog2_pixels_per_lane_variants_agree(px):
    frames = synth.numpy_frames("sat", 12, 16, 64, seed=7)
    eng = Engine(capi.MOG2)
    eng.set_option(capi.OPT_MOG2_PIXELS_PER_LANE, px)
    orc = pyoracle.Oracle(capi.MOG2)
    for f in frames:
        fg, bg = eng.process(f)
        ofg, obg = orc.process(f)
        assert np.array_equal(fg, ofg) and np.array_equal(bg, obg)
    check_mog2_state(eng, orc, 16 * 64)


@pytest.mark.parametrize("kw", [dict(alpha=0.005), dict(alpha=0.0), dict(alpha=-1.0), dict(alpha=1.0), dict(mog1_background_ratio=0.3),
                                dict(mog1_var_threshold=1.0), dict(mog1_noise_sigma=2.0), dict(enable_threshold=0)])
def test_mog1_param_variants(kw, golden_frames):
    p = _params(capi.MOG1, **kw)
    eng, orc, _ = run_pair(capi.MOG1, golden_frames[:12], params=p)
    check_mog1_state(eng, orc, golden_frames.shape[1] * golden_frames.shape[2])


def test_mog1_long_run():
    frames = synth.numpy_frames("surv", 100, 48, 80, seed=77)
    eng, orc, _ = run_pair(capi.MOG1, frames)
    check_mog1_state(eng, orc, 48 * 80)


@pytest.mark.parametrize("kw", [dict(learning_frames=3), dict(learning_frames=0), dict(learning_frames=-1), dict(threshold=5),
                                dict(alpha_learn=0.3, alpha_detection=0.01), dict(alpha_detection=1.0)])
def test_asbl_variants(kw, golden_frames):
    """learningFrames <= 0 skips the learning phase; after it only pixels whose median-filtered mask is 0 are updated."""
    run_pair(capi.ASBL, golden_frames[:12], params=_params(capi.ASBL, **kw))


@pytest.mark.parametrize("shape", [(48, 64), (37, 53), (5, 7), (3, 3), (1, 9), (130, 70)])
def test_asbl_ragged_sizes(shape):
    frames = synth.random_frames(8, shape[0], shape[1], 3, seed=shape[0])
    run_pair(capi.ASBL, frames, params=_params(capi.ASBL, learning_frames=2))


@pytest.mark.parametrize("shape,ch", [((40, 300), 3), ((9, 260), 3), ((70, 516), 3), ((33, 256), 3), ((20, 4), 3), ((41, 520), 1), ((2, 8), 1)])
def test_asbl_strip_geometries(shape, ch):
    """Rows of 4n pixels take the wave-per-strip table kernel (kernel_stencil.h: strips of 256 columns, a partial last strip, strips
    that end at the image border, fewer rows than a strip is tall); both phases, and a change of both alphas mid-run (the two tables
    are rebuilt)."""
    frames = synth.random_frames(10, shape[0], shape[1], ch, seed=shape[1])
    p = _params(capi.ASBL, learning_frames=4, threshold=12)
    eng = Engine(capi.ASBL, params=p)
    orc = pyoracle.Oracle(capi.ASBL, params=p)
    for t, f in enumerate(frames):
        if t == 6:
            p = _params(capi.ASBL, learning_frames=4, threshold=12, alpha_learn=0.2, alpha_detection=0.4)
            eng.set_params(p), orc.set_params(p)
        fg, bg = eng.process(f)
        ofg, obg = orc.process(f)
        assert np.array_equal(fg, ofg), "frame %d: %d mask pixels differ" % (t, int((fg != ofg).sum()))
        assert np.array_equal(bg, obg), "frame %d: %d background bytes differ" % (t, int((bg != obg).sum()))
    eng.close()


# ----------------------------------------------------------------------------- LBSP descriptors (pinned by the reference's own code)


@pytest.mark.parametrize("shape", [(48, 64), (37, 53), (5, 5), (4, 9), (80, 200), (17, 131)])
@pytest.mark.parametrize("ch", [1, 3])
def test_lbsp_seeded_vs_oracle(shape, ch):
    torch = _torch()
    from tracking_amd.engine import lbsp_describe_device
    rng = np.random.default_rng(shape[0] * 7 + ch)
    img = rng.integers(0, 256, shape + ((3,) if ch == 3 else ()), dtype=np.uint8)
    for rel, off in ((0.333, 0), (0.1, 3), (0.9, 0)):
        lut = pyoracle.lbsp_lut(rel, off, ch)
        want = pyoracle.lbsp_describe(img, lut)
        got = lbsp_describe_device(torch.from_numpy(img).cuda(), lut).cpu().numpy().view(np.uint16)
        assert np.array_equal(got, want)
        if pyoracle.ref_lbsp_available():  # only in the build container; the GPU box has the prebuilt .so
            assert np.array_equal(got, pyoracle.ref_lbsp_describe(img, lut))


@pytest.mark.parametrize("shape", [(48, 64), (37, 53), (3, 3), (1, 1), (130, 200)])
def test_mask_morphology_vs_oracle(shape):
    torch = _torch()
    from tracking_amd.engine import mask_morph_device, MORPH_ERODE, MORPH_DILATE, MORPH_MEDIAN
    rng = np.random.default_rng(shape[1])
    mask = np.where(rng.random(shape) < 0.45, 255, 0).astype(np.uint8)
    gray = rng.integers(0, 256, shape, dtype=np.uint8)
    d = torch.from_numpy(mask).cuda()
    for it in (1, 2, 3):
        assert np.array_equal(mask_morph_device(d, MORPH_ERODE, iterations=it).cpu().numpy(), pyoracle.erode3x3(mask, it))
        assert np.array_equal(mask_morph_device(d, MORPH_DILATE, iterations=it).cpu().numpy(), pyoracle.dilate3x3(mask, it))
    for k in (3, 9, 13):
        assert np.array_equal(mask_morph_device(d, MORPH_MEDIAN, ksize=k).cpu().numpy(), pyoracle.median_blur(mask, k))
    assert np.array_equal(mask_morph_device(torch.from_numpy(gray).cuda(), MORPH_MEDIAN, ksize=5).cpu().numpy(), pyoracle.median_blur(gray, 5))


# ----------------------------------------------------------------------------- SuBSENSE (BGR path)



def test_subsense_golden_frames(golden_frames):
    """96x80 < QVGA: 3x3 diffusion, median 9, no frame-level learning-rate scaling."""
    eng, orc, _ = run_pair(capi.SUBSENSE, golden_frames)
    check_subsense_state(eng, orc, golden_frames.shape[1], golden_frames.shape[2])


def test_subsense_flood_fill_finish_kernel_path(golden_frames, tmp_path):
    """BGS_SS_FLOOD_BATCH=0: no batch launches at all, SuBSENSE's hole filling is done entirely by ss_flood_finish_kernel (the
    path that otherwise only runs for masks the batch of relaxation launches does not converge on).  The knob is read once per process,
    hence the child process."""
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from gpu_helpers import run_pair, check_subsense_state, capi\n"
            "f = np.load(%r)['frames'][:10]\n"
            "eng, orc, _ = run_pair(capi.SUBSENSE, f)\n"
            "check_subsense_state(eng, orc, f.shape[1], f.shape[2])\n"
            "print('finish-kernel path OK')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)),
                                                   os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frames_96x80.npz"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BGS_SS_FLOOD_BATCH="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "finish-kernel path OK" in r.stdout, r.stdout + r.stderr


def test_round2_forms_of_the_round3_kernels_still_match(golden_frames):
    """Round 3 replaced several kernels and kept the earlier forms behind knobs that are read once per process (A/B builds, and the
    fallbacks for geometries the new forms do not take): SuBSENSE with the self updates in phase B, the tile flood fill and the
    LDS-count median; AdaptiveSelectiveBackgroundLearning through the LDS-tile kernel; LOBSTER's phase A with one pixel per lane in lock
    step (round 4 feeds the lanes from a queue).  One child process with all of them set:
    same masks, backgrounds and models as the oracle."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from gpu_helpers import run_pair, check_subsense_state, capi\n"
            "from tools import synth\n"
            "f = np.load(%r)['frames'][:12]\n"
            "eng, orc, _ = run_pair(capi.SUBSENSE, f)\n"
            "check_subsense_state(eng, orc, f.shape[1], f.shape[2])\n"
            "g = synth.numpy_frames('surv', 14, 240, 320, seed=5)\n"
            "eng, orc, _ = run_pair(capi.SUBSENSE, g)\n"
            "check_subsense_state(eng, orc, 240, 320)\n"
            "run_pair(capi.ASBL, synth.random_frames(8, 40, 300, 3, seed=2))\n"
            "from gpu_helpers import check_lobster_state\n"
            "eng, orc, _ = run_pair(capi.LOBSTER, g)\n"
            "check_lobster_state(eng, orc, 240, 320)\n"
            "print('round-2 forms OK')\n") % (os.path.dirname(here), here, os.path.join(here, "golden", "frames_96x80.npz"))
    env = dict(os.environ, BGS_SS_SELF_IN_A="0", BGS_SS_FLOOD_TILES="1", BGS_SS_MEDIAN_BITS="0", BGS_ASBL_TABLE="0", BGS_LOB_QUEUE="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "round-2 forms OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("knob", ["BGS_SS_FEEDBACK_SPLIT=1", "BGS_SS_QUEUE=1", "BGS_SS_IPASS_MIN=1 BGS_SS_FLOOD_WG1024=1 BGS_SS_B_LATE=1", "BGS_SS_IPASS_MIN=48 BGS_SS_REFILL=4"])
def test_subsense_round4_forms_behind_knobs_match_the_oracle(knob):
    """Round 4 built the two restructurings of phase A the verdict asked for - the rules behind the loop as a kernel of their own
    (BGS_SS_FEEDBACK_SPLIT=1) and the inter-LBSP tests worked off a per-wave list by whichever lane is free (BGS_SS_QUEUE=1) - measured
    both slower than the form that runs by default and kept them as A/B knobs.  Also behind knobs: the inter-LBSP passes
    taken at once (BGS_SS_IPASS_MIN=1, round 3's form; the default puts off a pass that fewer than 16 lanes would join) or put off
    until 48 lanes join, the flood strips in 1024-lane workgroups, phase B started behind the flood fill.  Knobs are read once per process: one
    child process each, same masks, backgrounds and whole model as the oracle incl. a scene cut (model reset) and a large frame."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from gpu_helpers import run_pair, check_subsense_state, capi\n"
            "from tools import synth\n"
            "f = np.load(%r)['frames'][:12]\n"
            "eng, orc, _ = run_pair(capi.SUBSENSE, f)\n"
            "check_subsense_state(eng, orc, f.shape[1], f.shape[2])\n"
            "a = synth.numpy_frames('surv', 20, 240, 320, seed=21) // 6\n"
            "b = 255 - synth.numpy_frames('surv', 8, 240, 320, seed=99) // 6\n"
            "eng, orc, _ = run_pair(capi.SUBSENSE, np.concatenate([a, b]))\n"
            "check_subsense_state(eng, orc, 240, 320)\n"
            "g = synth.numpy_frames('surv', 6, 360, 640, seed=7)\n"
            "eng, orc, _ = run_pair(capi.SUBSENSE, g)\n"
            "check_subsense_state(eng, orc, 360, 640)\n"
            "print('knob form OK')\n") % (os.path.dirname(here), here, os.path.join(here, "golden", "frames_96x80.npz"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **dict(kv.split("=") for kv in knob.split())), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "knob form OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("knob", ["BGS_SS_PARTS=4 BGS_SS_PART_MIN_PIXELS=1", "BGS_SS_PARTS=2 BGS_SS_PART_MIN_PIXELS=1 GPU_MAX_HW_QUEUES=8", "BGS_SS_PARTS=1 BGS_SS_PART_MIN_PIXELS=1",
                                  "BGS_SS_A_TOKEN=1 BGS_SS_PART_MIN_PIXELS=1"])
def test_subsense_batch_in_parts_and_ranges_on_their_own_streams_match_the_oracle(knob):
    """Round 4, engine_subsense.h: callers may drive stream ranges on HIP streams of their own (every call has its own pair of events on
    the side stream); behind knobs - measured, no gain, off by default - a large batch is cut into parts on HIP streams of the engine
    (BGS_SS_PARTS) and the phase A launches of all calls take turns (the "phase A token", BGS_SS_A_TOKEN=1) so that one part's phase B
    and post-processing run beside another part's phase A.  Five cameras with different scenes (one with a scene cut: refreshModel(0.1)),
    (a) as one batch call cut into 4 / 2 / 1 parts (5 streams: uneven parts), (b) as the ranges [0, 2) and [2, 5) on two HIP streams -
    every mask, background and whole model equals its own oracle's.  The size threshold is lowered so that 240 x 320 frames take the paths
    1080p batches take; knobs are read once per process: child processes."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from gpu_helpers import check_subsense_state, capi, Engine, pyoracle\n"
            "from tools import synth\n"
            "S, T, H, W = 5, 16, 240, 320\n"
            "clips = [synth.numpy_frames('surv', T, H, W, seed=30 + s) // (1 + s %% 3) for s in range(S)]\n"
            "clips[3][10:] = 255 - clips[3][10:]\n"
            "clips = np.stack(clips)\n"
            "for ranges in ([(0, 5)], [(0, 2), (2, 3)]):\n"
            "    eng = Engine(capi.SUBSENSE, n_streams=S); eng.set_geometry(H, W, 3)\n"
            "    orcs = [pyoracle.Oracle(capi.SUBSENSE) for _ in range(S)]\n"
            "    hs = [torch.cuda.Stream() for _ in ranges]\n"
            "    for t in range(T):\n"
            "        d_frames = torch.from_numpy(np.ascontiguousarray(clips[:, t])).cuda()\n"
            "        d_fg = torch.zeros((S, H, W), dtype=torch.uint8, device='cuda'); d_bg = torch.zeros((S, H, W, 3), dtype=torch.uint8, device='cuda')\n"
            "        torch.cuda.synchronize()\n"
            "        for (f, n), h in zip(ranges, hs):\n"
            "            if len(ranges) == 1: eng.process_batch_device(d_frames, d_fg, d_bg, None)\n"
            "            else: eng.process_batch_device(d_frames[f:f + n], d_fg[f:f + n], d_bg[f:f + n], None, hip_stream=h.cuda_stream, first=f, count=n)\n"
            "        torch.cuda.synchronize()\n"
            "        fg, bg = d_fg.cpu().numpy(), d_bg.cpu().numpy()\n"
            "        for s in range(S):\n"
            "            ofg, obg = orcs[s].process(clips[s, t])\n"
            "            assert np.array_equal(fg[s], ofg), (ranges, t, s, int((fg[s] != ofg).sum()))\n"
            "            assert np.array_equal(bg[s], obg), (ranges, t, s)\n"
            "    for s in range(S): check_subsense_state(eng, orcs[s], H, W, stream=s)\n"
            "    eng.close()\n"
            "print('parts and ranges OK')\n") % (os.path.dirname(here), here)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **dict(kv.split("=") for kv in knob.split())), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "parts and ranges OK" in r.stdout, r.stdout + r.stderr


def test_subsense_qvga_with_frame_level_block():
    """320x240 = QVGA: learning-rate scaling + auto model reset enabled; a scene cut at frame 12 triggers refreshModel(0.1)."""
    a = synth.numpy_frames("surv", 30, 240, 320, seed=21) // 6          # dark scene, long enough for ST (1/25) and LT (1/30) to part
    b = 255 - synth.numpy_frames("surv", 12, 240, 320, seed=99) // 6    # bright scene: large frame-level colour difference
    frames = np.concatenate([a, b])
    eng, orc, _ = run_pair(capi.SUBSENSE, frames)
    check_subsense_state(eng, orc, 240, 320)
    sc = eng.get_state("scalars", (7,), np.float64)
    assert sc[2] > 0, "the scene cut should have started a model-reset cooldown"


def test_subsense_large_frame_5x5_spread():
    """> 2x QVGA: 5x5 diffusion, larger median kernel (400x392 -> k = 11)."""
    frames = synth.numpy_frames("surv", 8, 392, 400, seed=5)
    eng, orc, _ = run_pair(capi.SUBSENSE, frames)
    check_subsense_state(eng, orc, 392, 400)


def test_subsense_modulo_by_multiplication_is_exact(golden_frames):
    """ss_feedback_kernel takes its five run-time `x % d` (which sample, which neighbour: BackgroundSubtractorSuBSENSE.cpp:508-551) as
    x - (mulhi(x, m[d]) >> (ceil(log2 d) - 1)) d with m from a 576-entry table (Granlund & Montgomery 1994, N = 31).  The table the
    DEVICE holds, against integer division: every d in 2..575, x over the edges of [0, 2^31) and 20 000 random draws each."""
    eng = Engine(capi.SUBSENSE)
    eng.process(golden_frames[0])
    m = eng.get_state("magic", (576,), np.uint32).astype(np.uint64)
    rng = np.random.default_rng(5)
    x = np.concatenate([np.array([0, 1, 2, 3, 2**31 - 1, 2**31 - 2, 2**30, 2**30 - 1, 2**30 + 1, 2**16, 2**16 - 1], np.uint64), rng.integers(0, 2**31, 20000).astype(np.uint64)])
    for d in range(2, 576):
        sh = np.uint64((d - 1).bit_length() - 1)  # 31 - clz(d - 1)
        q = ((x * m[d]) >> np.uint64(32)) >> sh
        assert np.array_equal(q, x // np.uint64(d)), d
        # multiples of d just below 2^31 (where an inexact multiplier would slip first)
        top = (np.uint64(2**31 - 1) // np.uint64(d)) * np.uint64(d)
        xs = np.array([top, top - np.uint64(1), min(top + np.uint64(d) - np.uint64(1), np.uint64(2**31 - 1))], np.uint64)
        assert np.array_equal(((xs * m[d]) >> np.uint64(32)) >> sh, xs // np.uint64(d)), d
    eng.close()


@pytest.mark.parametrize("shape", [(48, 64), (37, 53), (5, 5), (9, 131)])
def test_subsense_ragged_sizes(shape):
    frames = synth.random_frames(8, shape[0], shape[1], 3, seed=shape[1])
    eng, orc, _ = run_pair(capi.SUBSENSE, frames)
    check_subsense_state(eng, orc, shape[0], shape[1])


@pytest.mark.parametrize("kw", [dict(subsense_n_samples=20), dict(subsense_n_samples=3), dict(subsense_n_samples=5), dict(subsense_n_samples=63), dict(subsense_n_samples=100), dict(subsense_n_samples=130),  # 3 / 5 / 63: not whole batches of four; 100 / 130: past the 6-bit slot field of rounds 1-3
                                dict(subsense_n_required=3), dict(subsense_min_color_dist_threshold=15),
                                dict(subsense_desc_dist_threshold_offset=1), dict(lbsp_rel_threshold=0.2), dict(subsense_samples_for_moving_avgs=20)])
def test_subsense_param_variants(kw, golden_frames):
    p = _params(capi.SUBSENSE, **kw)
    eng, orc, _ = run_pair(capi.SUBSENSE, golden_frames[:10], params=p)
    check_subsense_state(eng, orc, golden_frames.shape[1], golden_frames.shape[2], nS=p.subsense_n_samples)


def test_subsense_streams_and_device_batch():
    torch = _torch()
    S, T, H, W = 3, 6, 40, 72
    clips = np.stack([synth.random_frames(T, H, W, 3, seed=300 + s) for s in range(S)])
    eng = Engine(capi.SUBSENSE, n_streams=S)
    eng.set_geometry(H, W, 3)
    orcs = [pyoracle.Oracle(capi.SUBSENSE) for _ in range(S)]
    for t in range(T):
        d_frames = torch.from_numpy(np.ascontiguousarray(clips[:, t])).cuda()
        d_fg = torch.empty((S, H, W), dtype=torch.uint8, device="cuda")
        d_bg = torch.empty((S, H, W, 3), dtype=torch.uint8, device="cuda")
        eng.process_batch_device(d_frames, d_fg, d_bg, None)
        torch.cuda.synchronize()
        for s in range(S):
            ofg, obg = orcs[s].process(clips[s, t])
            assert np.array_equal(d_fg[s].cpu().numpy(), ofg), (t, s)
            assert np.array_equal(d_bg[s].cpu().numpy(), obg), (t, s)
    for s in range(S):
        check_subsense_state(eng, orcs[s], H, W, stream=s)


def test_subsense_grayscale_path(golden_gray):
    """CV_8UC1 input: the 1-channel branch of operator() (BackgroundSubtractorSuBSENSE.cpp:306-434), LUT / 3, halved colour threshold."""
    eng, orc, _ = run_pair(capi.SUBSENSE, golden_gray)
    check_subsense_state(eng, orc, golden_gray.shape[1], golden_gray.shape[2], C=1)


def test_subsense_grayscale_qvga_scene_cut():
    a = synth.numpy_frames("surv", 30, 240, 320, seed=8)[..., 1] // 6
    b = 255 - synth.numpy_frames("surv", 12, 240, 320, seed=9)[..., 1] // 6
    frames = np.ascontiguousarray(np.concatenate([a, b]))
    eng, orc, _ = run_pair(capi.SUBSENSE, frames)
    check_subsense_state(eng, orc, 240, 320, C=1)


@pytest.mark.parametrize("shape,ch", [((243, 325), 3), ((250, 333), 1)])
def test_subsense_sizes_that_are_not_multiples_of_8(shape, ch):
    """>= QVGA and not a multiple of 8 (the reference takes any size: BackgroundSubtractorSuBSENSE.cpp:153 down-samples to
    width/8 x height/8 with cv::resize INTER_AREA, whose general path has fractional cell weights): learning-rate scaling + auto model
    reset enabled; a scene cut triggers refreshModel(0.1).  The whole model must equal the oracle, frame-level scalars included."""
    H, W = shape
    a = synth.numpy_frames("surv", 30, H, W, seed=21) // 6
    b = 255 - synth.numpy_frames("surv", 10, H, W, seed=99) // 6
    frames = np.concatenate([a, b])
    if ch == 1:
        frames = np.ascontiguousarray(frames[..., 1])
    eng, orc, _ = run_pair(capi.SUBSENSE, frames)
    check_subsense_state(eng, orc, H, W, C=ch)
    sc = eng.get_state("scalars", (7,), np.float64)
    if ch == 3:  # (the 1-channel measure is |ST - LT| / 2, :664: this cut stays below its threshold - the scalars still equal the oracle's)
        assert sc[2] > 0, "the scene cut should have started a model-reset cooldown"


def test_subsense_aged_model_with_scene_cut_640x360():
    """The states the performance figures are quoted on, oracle-checked in the suite (round-2 verdict): a model aged 60 frames at
    640 x 360 (> 2 x QVGA: 5x5 diffusion, learning-rate scaling, update rates settled on the static part), a scene cut at frame 44 with
    refreshModel(0.1), and the frames after it.  Masks every frame, the whole model at the end."""
    a = synth.numpy_frames("surv", 44, 360, 640, seed=61) // 5
    b = 255 - synth.numpy_frames("surv", 16, 360, 640, seed=62) // 5
    frames = np.concatenate([a, b])
    eng, orc, _ = run_pair(capi.SUBSENSE, frames, want_bg=False)
    check_subsense_state(eng, orc, 360, 640)
    sc = eng.get_state("scalars", (7,), np.float64)
    assert sc[0] == 60 and sc[2] > 0, sc  # 60 frames seen, the cut started a model-reset cooldown


def test_subsense_854x480_not_a_multiple_of_8():
    """854 x 480 (the verdict's example; 854 = 8 * 106 + 6): 5x5 diffusion, median 13, the fractional down-sampling on the x axis only."""
    a = synth.numpy_frames("surv", 7, 480, 854, seed=3) // 5
    b = 255 - synth.numpy_frames("surv", 3, 480, 854, seed=4) // 5
    frames = np.concatenate([a, b])
    eng, orc, _ = run_pair(capi.SUBSENSE, frames)
    check_subsense_state(eng, orc, 480, 854)


@pytest.mark.parametrize("shape", [(64, 64), (37, 53), (1, 1), (3, 200), (130, 257), (200, 70), (300, 520), (1100, 130), (2100, 70), (4200, 64)])
def test_floodfill_from_origin_vs_oracle(shape):
    """cv::floodFill(mask, Point(0,0), 255): mazes with long snaking corridors, enclosed holes, origin on either value,
    sizes that are not multiples of the 64x64 bit-packed tile.  The serpentine walls make the fill cross tile borders far more
    often than the fixed batch of relaxation launches covers (kSsFloodBatch x kSsFloodRounds tile steps), so ss_flood_finish_kernel does the rest
    (at 300x520: 45 tiles over the finish kernel's 16 waves).  Heights above 1024 give a strip workgroup's waves several tiles
    each (ss_flood_strip_kernel), above 4096 rows the tile kernel runs instead."""
    torch = _torch()
    from tracking_amd.engine import mask_morph_device, MORPH_FLOODFILL_ORIGIN, MORPH_MEDIAN_BINARY
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    for density, origin in ((0.3, 0), (0.45, 0), (0.6, 255), (0.0, 0)):
        m = np.where(rng.random(shape) < density, 255, 0).astype(np.uint8)
        if shape[0] > 8 and shape[1] > 8:  # a serpentine wall: forces a long path
            m[4:-4:8, :-3] = 255
            m[8:-4:8, 3:] = 255
        m[0, 0] = origin
        got = mask_morph_device(torch.from_numpy(m).cuda(), MORPH_FLOODFILL_ORIGIN).cpu().numpy()
        assert np.array_equal(got, pyoracle.floodfill_from_origin(m, 255)), (density, origin)
    b = np.where(rng.random(shape) < 0.5, 255, 0).astype(np.uint8)
    for k in (3, 9, 13):
        assert np.array_equal(mask_morph_device(torch.from_numpy(b).cuda(), MORPH_MEDIAN_BINARY, ksize=k).cpu().numpy(), pyoracle.median_blur(b, k))


@pytest.mark.parametrize("weighted", [1, 0])
def test_wmm_every_byte_triple_matches_oracle(weighted):
    """wmm_kernel takes the background byte from integer arithmetic except on exact ties of the weighted mean (M = 5 b0 + 3 b1 + 2 b2
    ending in 5), which go through the float pipeline (kernel_pointwise.h).  Every one of the 2^24 byte triples, background image and
    unthresholded mask, against the oracle's float pipeline; weighted and unweighted."""
    p = np.arange(1 << 24, dtype=np.uint32).reshape(4096, 4096)
    frames = [(p >> 16).astype(np.uint8), ((p >> 8) & 255).astype(np.uint8), (p & 255).astype(np.uint8)]  # t-2, t-1, t
    prm = _params(capi.WMM, enable_threshold=0, enable_weight=weighted)
    eng, orc = Engine(capi.WMM, params=prm), pyoracle.Oracle(capi.WMM, params=prm)
    for f in frames:
        fg, bg = eng.process(f)
        ofg, obg = orc.process(f)
    for name, got, want in (("background", bg, obg), ("mask", fg, ofg)):
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "%s: %d of 2^24 triples differ, first at %s: %d vs %d" % (name, bad.size, hex(int(bad[0])), int(got.flat[bad[0]]), int(want.flat[bad[0]]))
    eng.close()


def test_wmv_every_byte_triple_matches_oracle():
    """wmv_kernel takes the byte of a moving pixel from integer arithmetic unless it lies near a rounding boundary of the reference's
    float pipeline (kernel_pointwise.h: wmv_fast_byte).  Every one of the 2^24 (current, previous, before-previous) byte triples, as
    three 4096 x 4096 one-channel frames with the threshold off (the mask then IS the byte), against the oracle's float pipeline."""
    p = np.arange(1 << 24, dtype=np.uint32).reshape(4096, 4096)
    frames = [(p >> 16).astype(np.uint8), ((p >> 8) & 255).astype(np.uint8), (p & 255).astype(np.uint8)]  # t-2, t-1, t
    prm = _params(capi.WMV, enable_threshold=0)
    eng, orc = Engine(capi.WMV, params=prm), pyoracle.Oracle(capi.WMV, params=prm)
    for f in frames:
        fg, _ = eng.process(f)
        ofg, _ = orc.process(f)
    bad = np.flatnonzero(fg != ofg)
    assert bad.size == 0, "%d of 2^24 triples differ, first at %s: %d vs %d" % (bad.size, hex(int(bad[0])), int(fg.flat[bad[0]]), int(ofg.flat[bad[0]]))
    eng.close()


@pytest.mark.parametrize("thr", [0, 1, 7, 15, 60, 128, 250])
def test_wmv_mask_band_around_threshold(thr):
    """Threshold on: wmv_kernel settles a moving pixel's mask from the integer bytes unless their gray value is within 2 of the
    threshold, and sends only those pixels' bytes through the float pipeline.  Full-range random frames put ~2 % of the pixels
    into that band at every threshold; 3 channels and 1."""
    rng = np.random.default_rng(1000 + thr)
    frames = rng.integers(0, 256, (6, 96, 256, 3), dtype=np.uint8)
    frames[3:] = (frames[3:].astype(np.int32) // 4 + frames[2].astype(np.int32) * 3 // 4).astype(np.uint8)  # smaller steps too: low gray values
    run_pair(capi.WMV, frames, params=_params(capi.WMV, threshold=thr))
    run_pair(capi.WMV, frames[:, :, :, 1], params=_params(capi.WMV, threshold=thr))


@pytest.mark.parametrize("thr", [1, 2, 15, 40, 127, 200])
def test_wmv_quiet_pixel_shortcut_is_exact(thr):
    """wmv_kernel skips the float pipeline when every channel's temporal range is < 2*thr (proof in kernel_pointwise.h).
    Probe the boundary: ranges 2*thr-2 .. 2*thr+1 in every 0.5|0.5-like split, all three history slots, plus random data."""
    rng = np.random.default_rng(thr)
    H, W = 32, 64
    base = rng.integers(0, max(1, 255 - 2 * thr - 2), (H, W, 3))
    frames = []
    for t in range(9):
        delta = rng.integers(2 * thr - 2, 2 * thr + 2, (H, W, 3))
        on = rng.random((H, W, 3)) < 0.5
        frames.append(np.clip(base + np.where(on, delta, 0), 0, 255).astype(np.uint8))
    frames = np.stack(frames)
    run_pair(capi.WMV, frames, params=_params(capi.WMV, threshold=thr))
    run_pair(capi.WMV, frames[:, :, :, 0], params=_params(capi.WMV, threshold=thr))
    run_pair(capi.WMV, frames, params=_params(capi.WMV, threshold=thr, enable_weight=0))


# ----------------------------------------------------------------------------- BASELINE configs[2], configs[3] at full size


def test_gmg_through_training_and_operation(golden_frames):
    """48 frames: 20 training frames (histogram build-up, normalisation on frame 19), then decisions + move-to-front updates."""
    frames = np.concatenate([golden_frames, golden_frames[::-1]])
    eng, orc, _ = run_pair(capi.GMG, frames)
    check_state("GMG", eng, orc, frames.shape[1] * frames.shape[2])


@pytest.mark.parametrize("kw", [dict(gmg_init_frames=3), dict(gmg_max_features=4, gmg_init_frames=5), dict(gmg_quantization_levels=64, gmg_init_frames=4),
                                dict(gmg_smoothing_radius=0, gmg_init_frames=4), dict(gmg_smoothing_radius=3, gmg_decision_threshold=0.9, gmg_init_frames=4),
                                dict(gmg_learning_rate=0.2, gmg_background_prior=0.5, gmg_init_frames=2), dict(gmg_update_background_model=0, gmg_init_frames=1)])
def test_gmg_variants(kw, golden_frames):
    """small maxFeatures forces the drop-the-oldest path; 64 quantisation levels makes new features (and appends) frequent."""
    p = _params(capi.GMG, **kw)
    eng = Engine(capi.GMG, params=p)
    orc = pyoracle.Oracle(capi.GMG, params=p)
    for t, f in enumerate(golden_frames[:16]):
        fg, bg = eng.process(f)
        ofg, obg = orc.process(f)
        assert bg is None and obg is None and np.array_equal(fg, ofg), t
    n = golden_frames.shape[1] * golden_frames.shape[2]
    assert np.array_equal(eng.get_state("nfeatures", (n,), np.int32), orc.get_state("nfeatures", (n,), np.int32))
    F = 64
    a, b = eng.get_state("weights", (p.gmg_max_features, n), np.float32), orc.get_state("weights", (p.gmg_max_features, n), np.float32)
    assert float(np.max(np.abs(a - b))) <= STATE_TOL
    assert np.array_equal(eng.get_state("colors", (p.gmg_max_features, n), np.int32), orc.get_state("colors", (p.gmg_max_features, n), np.int32))


def test_gmg_gray_and_ragged(golden_gray):
    run_pair(capi.GMG, np.concatenate([golden_gray, golden_gray]), params=_params(capi.GMG, gmg_init_frames=6))
    frames = synth.random_frames(12, 37, 53, 3, seed=4)
    run_pair(capi.GMG, frames, params=_params(capi.GMG, gmg_init_frames=5))


@pytest.mark.parametrize("shape", [(64, 64), (37, 53), (1, 1), (1, 300), (300, 1), (130, 257), (240, 320)])
@pytest.mark.parametrize("connectivity", [8, 4])
def test_connected_components_vs_oracle(shape, connectivity):
    """N1: labels (root per pixel), boxes sorted by root, areas, count - identical to the oracle's flood-fill labelling."""
    torch = _torch()
    from tracking_amd.engine import mask_components_device
    rng = np.random.default_rng(shape[0] * 7919 + shape[1] + connectivity)
    for tag, m in _cc_masks(shape, rng):
        want_l, want_b, want_n = pyoracle.components(m, connectivity)
        labels, boxes, n = mask_components_device(torch.from_numpy(m).cuda(), connectivity, max_boxes=shape[0] * shape[1] + 1)
        assert n == want_n, (tag, n, want_n)
        assert np.array_equal(labels.cpu().numpy(), want_l), tag
        got = boxes.cpu().numpy()
        want = np.stack([want_b[f] for f in ("x", "y", "w", "h", "area", "root")], axis=1) if want_n else np.zeros((0, 6), np.int32)
        assert np.array_equal(got, want), tag


def test_connected_components_truncation_and_full_size():
    """max_boxes smaller than the component count: the count is still the total, the first boxes are still exact;
    1080p mask from the SuBSENSE-like blob generator; no label image requested."""
    torch = _torch()
    from tracking_amd.engine import mask_components_device
    rng = np.random.default_rng(5)
    m = np.where(rng.random((96, 160)) < 0.2, 255, 0).astype(np.uint8)
    want_l, want_b, want_n = pyoracle.components(m, 8)
    assert want_n > 40
    labels, boxes, n = mask_components_device(torch.from_numpy(m).cuda(), 8, max_boxes=40, want_labels=False)
    assert labels is None and n == want_n and boxes.shape[0] == 40
    assert np.array_equal(boxes.cpu().numpy()[:, 5], want_b["root"][:40]) and np.array_equal(boxes.cpu().numpy()[:, 4], want_b["area"][:40])
    big = np.zeros((1080, 1920), np.uint8)
    for _ in range(60):
        y, x = rng.integers(0, 1000), rng.integers(0, 1800)
        big[y:y + rng.integers(5, 80), x:x + rng.integers(5, 120)] = 255
    big[rng.random(big.shape) < 0.001] = 255
    want_l, want_b, want_n = pyoracle.components(big, 8)
    labels, boxes, n = mask_components_device(torch.from_numpy(big).cuda(), 8, max_boxes=8192)
    assert n == want_n and np.array_equal(labels.cpu().numpy(), want_l)
    assert np.array_equal(boxes.cpu().numpy(), np.stack([want_b[f] for f in ("x", "y", "w", "h", "area", "root")], axis=1))


@pytest.mark.parametrize("kw", [dict(dp_gaussians=1), dict(dp_gaussians=2), dict(dp_gaussians=5), dict(dp_alpha=0.2), dict(dp_alpha=0.6, dp_gaussians=4),
                                dict(dp_threshold=2.0), dict(dp_threshold=400.0, dp_alpha=0.05)])
@pytest.mark.parametrize("name", ["DPZivkovicAGMMBGS", "DPGrimsonGMMBGS"])
def test_dp_gmm_parameter_variants(name, kw):
    """MaxModes 1..5 (template instances), large alpha (weights fall under the prune limit: the mode count shrinks inside
    the update loop, ZivkovicAGMM.cpp:240-245), tight and loose thresholds."""
    frames = synth.random_frames(30, 24, 40, 3, seed=len(name) + int(kw.get("dp_gaussians", 0)))
    frames[10:20] = frames[:10]  # repeated content: existing modes get matched, not only created
    p = _params(ALGOS[name], **kw)
    eng, orc, _ = run_pair(ALGOS[name], frames, params=p)
    check_dp_state(name, eng, orc, 24 * 40, K=p.dp_gaussians)


@pytest.mark.parametrize("name,kw", [("DPWrenGABGS", dict(dp_alpha=0.3)), ("DPWrenGABGS", dict(dp_threshold=1.0)), ("DPMeanBGS", dict(dp_alpha=0.9)),
                                     ("DPMeanBGS", dict(dp_threshold=100.0, dp_alpha=0.5)), ("DPAdaptiveMedianBGS", dict(dp_sampling_rate=1)),
                                     ("DPAdaptiveMedianBGS", dict(dp_sampling_rate=2, dp_threshold=3.0)), ("DPAdaptiveMedianBGS", dict(dp_sampling_rate=3, dp_threshold=0.5))])
def test_dp_simple_models_parameter_variants(name, kw, golden_frames):
    eng, orc, _ = run_pair(ALGOS[name], golden_frames, params=_params(ALGOS[name], **kw))
    check_dp_state(name, eng, orc, golden_frames.shape[1] * golden_frames.shape[2])


def test_dp_models_reject_single_channel(golden_gray):
    for name in DP_NAMES:
        with pytest.raises(capi.BgsError) as ei:
            Engine(ALGOS[name]).process(golden_gray[0])
        assert ei.value.code == capi.ERR_UNSUPPORTED
        with pytest.raises(RuntimeError):
            pyoracle.Oracle(ALGOS[name]).process(golden_gray[0])


@pytest.mark.parametrize("shape", [(37, 53), (5, 5), (6, 70), (130, 67)])
def test_lobster_ragged_sizes(shape):
    frames = synth.random_frames(8, shape[0], shape[1], 3, seed=shape[0] + shape[1])
    frames[4:] = frames[:4]  # repeated content so that part of the image is background and posts update requests
    eng, orc, _ = run_pair(capi.LOBSTER, frames)
    check_lobster_state(eng, orc, shape[0], shape[1])


@pytest.mark.parametrize("kw", [dict(subsense_n_samples=8, subsense_n_required=1), dict(subsense_n_samples=20, subsense_n_required=4), dict(subsense_n_samples=100), dict(subsense_min_color_dist_threshold=12),
                                dict(subsense_desc_dist_threshold_offset=1), dict(lbsp_rel_threshold=0.2, lbsp_threshold_offset=6)])
def test_lobster_param_variants(kw, golden_frames):
    p = _params(capi.LOBSTER, **kw)
    eng, orc, _ = run_pair(capi.LOBSTER, golden_frames[:14], params=p)
    check_lobster_state(eng, orc, golden_frames.shape[1], golden_frames.shape[2], nS=p.subsense_n_samples)


def test_lobster_grayscale_and_streams(golden_gray):
    """1-channel path (BackgroundSubtractorLOBSTER.cpp:183-223: thresholds / 2, LUT / 2) and 3 interleaved streams through the device batch."""
    eng, orc, _ = run_pair(capi.LOBSTER, golden_gray)
    check_lobster_state(eng, orc, golden_gray.shape[1], golden_gray.shape[2], C=1)
    torch = _torch()
    S, T, H, W = 3, 7, 40, 72
    clips = np.stack([synth.s_smooth(T, H, W, seed=50 + s, device="cpu").numpy() for s in range(S)])
    eng = Engine(capi.LOBSTER, n_streams=S)
    eng.set_geometry(H, W, 3)
    orcs = [pyoracle.Oracle(capi.LOBSTER) for _ in range(S)]
    for t in range(T):
        d_frames = torch.from_numpy(np.ascontiguousarray(clips[:, t])).cuda()
        d_fg = torch.empty((S, H, W), dtype=torch.uint8, device="cuda")
        d_bg = torch.empty((S, H, W, 3), dtype=torch.uint8, device="cuda")
        eng.process_batch_device(d_frames, d_fg, d_bg, None)
        for s in range(S):
            ofg, obg = orcs[s].process(clips[s, t])
            assert np.array_equal(d_fg[s].cpu().numpy(), ofg) and np.array_equal(d_bg[s].cpu().numpy(), obg), (t, s)
    for s in range(S):
        check_lobster_state(eng, orcs[s], H, W, stream=s)


@pytest.mark.parametrize("connectivity", [8, 4])
def test_connected_components_batch(connectivity):
    """Stack of masks in one call: no component may leak across an image boundary (first/last rows are made busy on purpose),
    boxes come out grouped per image in root order, offsets are the prefix sums of the per-image counts; one image is empty."""
    torch = _torch()
    from tracking_amd.engine import mask_components_batch_device
    rng = np.random.default_rng(connectivity)
    S, H, W = 5, 37, 70
    masks = np.where(rng.random((S, H, W)) < 0.35, 255, 0).astype(np.uint8)
    masks[:, 0, :] = 255
    masks[:, -1, ::2] = 255
    masks[3] = 0
    labels, boxes, off = mask_components_batch_device(torch.from_numpy(masks).cuda(), connectivity, max_boxes=S * H * W, want_labels=True)
    got_b, got_l, off = boxes.cpu().numpy(), labels.cpu().numpy(), off.numpy()
    assert off[0] == 0 and len(off) == S + 1
    for s in range(S):
        want_l, want_b, want_n = pyoracle.components(masks[s], connectivity)
        assert off[s + 1] - off[s] == want_n, s
        assert np.array_equal(got_l[s], want_l), s
        want = np.stack([want_b[f] for f in ("x", "y", "w", "h", "area", "root")], axis=1) if want_n else np.zeros((0, 6), np.int32)
        assert np.array_equal(got_b[off[s]:off[s + 1]], want), s


def test_lbsp_batch_equals_single_images():
    """bgs_lbsp_describe_batch_device: a stack of frames in one launch gives the descriptors of each frame alone (and the oracle's)."""
    torch = _torch()
    from tracking_amd.engine import lbsp_describe_batch_device, lbsp_describe_device
    lut = pyoracle.lbsp_lut(0.333, 0, 3)
    imgs = synth.random_frames(3, 46, 68, 3, seed=77)
    d = torch.from_numpy(imgs).cuda()
    got = lbsp_describe_batch_device(d, lut).cpu().numpy().view(np.uint16)
    for k in range(3):
        assert np.array_equal(got[k], lbsp_describe_device(d[k], lut).cpu().numpy().view(np.uint16))
        assert np.array_equal(got[k], pyoracle.lbsp_describe(imgs[k], lut))


def test_model_sizing_parameters_are_frozen_after_the_first_frame(golden_frames):
    """The reference hands these values to its model once (SuBSENSE.cpp:27-36, DP*BGS.cpp `if(firstTime)`): a later bgs_set_params
    with a different sample / gaussian count must neither resize nor overrun the model - it is ignored, as in the reference."""
    for algo, field, big in ((capi.LOBSTER, "subsense_n_samples", 60), (capi.SUBSENSE, "subsense_n_samples", 63), (capi.DP_ZIVKOVIC_AGMM, "dp_gaussians", 5),
                             (capi.DP_GRIMSON_GMM, "dp_gaussians", 5)):
        p = _params(algo, **{field: 2})
        if field == "subsense_n_samples":
            p.subsense_n_required = 1
        eng, orc = Engine(algo, params=p), pyoracle.Oracle(algo, params=p)
        for t, f in enumerate(golden_frames[:6]):
            if t == 2:
                q = _params(algo, **{field: big})
                eng.set_params(q)  # ignored for the running model
            fg, bg = eng.process(f)
            ofg, obg = orc.process(f)
            assert np.array_equal(fg, ofg), (algo, t)


@pytest.mark.parametrize("algo", [capi.MOG2, capi.SUBSENSE, capi.DP_ZIVKOVIC_AGMM, capi.ABL])
def test_disjoint_stream_ranges_in_flight_on_two_hip_streams(algo):
    """bgs_process_range_device for streams [0, 2) on one HIP stream and [2, 4) on another, enqueued back to back without any
    synchronisation in between, frame after frame: every stream must equal its own oracle.  (What the header allows: ranges in
    flight at once must be disjoint; per-range state - SuBSENSE's flood-fill flags, frame counters, first-frame initialisation on
    the launch stream - must not be shared.)"""
    torch = _torch()
    H, W, S, T = 48, 64, 4, 6
    clips = np.stack([synth.random_frames(T, H, W, 3, seed=300 + s) for s in range(S)])  # [S][T][H][W][3]
    dev = torch.from_numpy(clips).cuda()
    eng = Engine(algo, n_streams=S)
    eng.set_geometry(H, W, 3)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    fg = torch.empty((T, S, H, W), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for t in range(T):
        fa, fb = dev[0:2, t].contiguous(), dev[2:4, t].contiguous()
        torch.cuda.synchronize()  # the frame copies above ran on torch's default stream
        eng.process_batch_device(fa, fg[t, 0:2], None, None, hip_stream=s1.cuda_stream, first=0, count=2)
        eng.process_batch_device(fb, fg[t, 2:4], None, None, hip_stream=s2.cuda_stream, first=2, count=2)
    torch.cuda.synchronize()
    got = fg.cpu().numpy()
    for s in range(S):
        orc = pyoracle.Oracle(algo)
        for t in range(T):
            ofg, _ = orc.process(clips[s, t], want_bg=False)
            assert np.array_equal(got[t, s], ofg), (s, t, int((got[t, s] != ofg).sum()))
    eng.close()


def test_engines_release_their_device_memory():
    """Create -> geometry -> a few frames (host and device path) -> destroy, every class, three times over: the device's free memory
    returns to where it started (every buffer an engine allocates - models, bit planes, staging, probe candidates - is freed)."""
    torch = _torch()
    H, W = 96, 128
    frames = synth.random_frames(4, H, W, 3, seed=1)
    d = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
    algos = list(ALGOS.values()) + [capi.SUBSENSE, capi.LOBSTER]

    def cycle():
        for algo in algos:
            e = Engine(algo, n_streams=2)
            e.set_geometry(H, W, 3)
            fg = torch.zeros((2, H, W), dtype=torch.uint8, device="cuda")
            for t in range(3):
                e.process_batch_device(torch.stack([d[t], d[t + 1]]).contiguous(), fg)
            e.process_clip_device(torch.stack([torch.stack([d[0], d[1]]), torch.stack([d[2], d[3]])]).contiguous(), 2, None)
            torch.cuda.synchronize()
            e.close()
            h = Engine(algo)
            h.process(frames[0])
            h.process(frames[1])
            h.close()

    cycle()  # first pass: one-time allocations of the runtime itself
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(3):
        cycle()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (8 << 20), "device memory shrank by %d bytes over three create/destroy cycles" % (free0 - free1)


@pytest.mark.parametrize("name", ["MixtureOfGaussianV2BGS", "FrameDifferenceBGS", "AdaptiveBackgroundLearning"])
def test_host_path_with_registered_caller_buffers(name, golden_frames):
    """BGS_OPT_HOST_REGISTER: the caller keeps ONE frame buffer, ONE mask and ONE background image allocated (what OpenCV's capture loop
    and a cv::Mat member are); from the second call on they are page-locked and read / written by the DMA engine in place.  Same
    results as the staged path; switching buffers mid-run drops the registration and takes up the new one."""
    algo = ALGOS[name]
    eng, orc = Engine(algo), pyoracle.Oracle(algo)
    eng.set_option(capi.OPT_HOST_REGISTER, 7)
    H, W = golden_frames.shape[1:3]
    bufs = [(np.empty((H, W, 3), np.uint8), np.full((H, W), 9, np.uint8), np.full((H, W, 3), 9, np.uint8)) for _ in range(2)]
    for t, f in enumerate(golden_frames[:16]):
        frame, fg, bg = bufs[0 if t < 9 else 1]  # second set of buffers from frame 9 on
        frame[...] = f
        fg[...] = 9
        flags = eng.process_into(frame, fg, bg)
        ofg, obg = orc.process(f)
        assert bool(flags & capi.FG_VALID) == (ofg is not None) and bool(flags & capi.BG_VALID) == (obg is not None), (name, t)
        if ofg is not None:
            assert np.array_equal(fg, ofg), (name, t)
        else:
            assert (fg == 9).all(), (name, t)
        if obg is not None:
            assert np.array_equal(bg, obg), (name, t)
    eng.set_option(capi.OPT_HOST_REGISTER, 0)  # back to staging: unregisters
    fg2, _ = eng.process(golden_frames[16])
    ofg, _ = orc.process(golden_frames[16])
    assert np.array_equal(fg2, ofg)
    eng.close()


@pytest.mark.parametrize("name", ["MixtureOfGaussianV2BGS", "WeightedMovingVarianceBGS", "SuBSENSEBGS", "AdaptiveSelectiveBackgroundLearning"])
@pytest.mark.parametrize("register", [0, 7])
def test_submit_wait_several_cameras_overlap(name, register, golden_frames):
    """bgs_submit / bgs_wait: four cameras on the host path, every camera's frame t queued before any is collected (their uploads,
    kernels and downloads overlap on per-camera lanes), one camera lagging a frame behind, one synchronous bgs_process in between.
    Every camera equals its own oracle; with BGS_OPT_HOST_REGISTER the per-camera buffers are used in place."""
    algo = dict(ALGOS, SuBSENSEBGS=capi.SUBSENSE)[name]
    S, T = 4, 9
    H, W = golden_frames.shape[1:3]
    clips = [np.ascontiguousarray(golden_frames[2 * s:2 * s + T]) for s in range(S)]
    eng = Engine(algo, n_streams=S)
    eng.set_option(capi.OPT_HOST_REGISTER, register)
    orcs = [pyoracle.Oracle(algo) for _ in range(S)]
    bg_c = 1 if algo == capi.ASBL else 3
    frames = [np.empty((H, W, 3), np.uint8) for _ in range(S)]  # one frame / mask / background buffer per camera, kept allocated
    fgs = [np.full((H, W), 9, np.uint8) for _ in range(S)]
    bgs_ = [np.full((H, W, bg_c), 9, np.uint8) for _ in range(S)]
    fed = [0] * S
    for t in range(T):
        cams = [s for s in range(S) if not (s == 2 and t == 0)]  # camera 2 comes up one step late
        for s in cams:
            frames[s][...] = clips[s][fed[s]]
            fgs[s][...] = 9
            eng.submit(frames[s], fgs[s], bgs_[s], stream=s)
        if t == 4:  # a synchronous call on a camera with a submission in flight collects that one first, then runs
            s = 1
            fl = eng.wait(stream=s)
            ofg, obg = orcs[s].process(clips[s][fed[s]])
            assert bool(fl & capi.FG_VALID) == (ofg is not None)
            if ofg is not None:
                assert np.array_equal(fgs[s], ofg)
            fed[s] += 1
            fg2, _ = eng.process(clips[s][fed[s]], stream=s)
            ofg, _ = orcs[s].process(clips[s][fed[s]])
            assert (fg2 is None) == (ofg is None) and (ofg is None or np.array_equal(fg2, ofg))
            fed[s] += 1
            cams = [c for c in cams if c != s]
        for s in cams:
            fl = eng.wait(stream=s)
            ofg, obg = orcs[s].process(clips[s][fed[s]])
            assert bool(fl & capi.FG_VALID) == (ofg is not None) and bool(fl & capi.BG_VALID) == (obg is not None), (name, t, s, fl)
            if ofg is not None:
                assert np.array_equal(fgs[s], ofg), (name, t, s)
            else:
                assert (fgs[s] == 9).all(), (name, t, s)
            if obg is not None:
                assert np.array_equal(bgs_[s].reshape(obg.shape), obg), (name, t, s)
            fed[s] += 1
        if fed[1] >= T - 1:
            break
    with pytest.raises(capi.BgsError):
        eng.submit(frames[0], fgs[0], None, stream=0)
        eng.submit(frames[0], fgs[0], None, stream=0)  # second submission while the first is in flight
    eng.wait(stream=0)
    eng.close()
