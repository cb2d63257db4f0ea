"""The GPU parity tests that matter most, collected FIRST (file name sorts before every other test module) so that a late
failure elsewhere can never hide them: BASELINE.json's configurations at full size, the bench.py geometry itself
(32 x 1080p streams in one launch), the MOG2 data-dependent traffic levels, >4 GB models, one LOBSTER and one dp/ long clip.

Bar (BASELINE.json north_star): uint8 masks / backgrounds bit-exact, float model state within 1e-4.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import pyoracle
from tools import synth
from tracking_amd import Engine, capi

from gpu_helpers import *  # noqa: F401,F403
from gpu_helpers import _params, _torch

pytestmark = pytest.mark.gpu


def test_full_size_1080p_mog2_sampled_parity():
    """BASELINE.json config 2 at full size: 1920x1080, S_sat frames generated in HBM.  MOG2 is pointwise, so the oracle
    replays the exact same 20-frame history on a 64k-pixel random sample and must agree bit-for-bit there; the whole
    frame is covered by size-independent invariants (weights sorted & normalised, variance clamped, 1 <= nmodes <= 5)."""
    torch = _torch()
    H, W, T = 1080, 1920, 20
    frames = synth.s_sat(T, H, W, seed=1234, device="cuda")
    eng = Engine(capi.MOG2)
    eng.set_geometry(H, W, 3)
    d_fg = torch.empty((T, H, W), dtype=torch.uint8, device="cuda")
    for t in range(T):
        eng.process_batch_device(frames[t:t + 1], d_fg[t:t + 1], None, None)
    torch.cuda.synchronize()
    rng = np.random.default_rng(99)
    idx = rng.choice(H * W, 65536, replace=False)
    idx.sort()
    sample = frames.reshape(T, H * W, 3)[:, torch.from_numpy(idx).cuda()].cpu().numpy().reshape(T, 256, 256, 3)
    fg_s = d_fg.reshape(T, H * W)[:, torch.from_numpy(idx).cuda()].cpu().numpy().reshape(T, 256, 256)
    orc = pyoracle.Oracle(capi.MOG2)
    for t in range(T):
        ofg, _ = orc.process(sample[t], want_bg=False)
        assert np.array_equal(fg_s[t], ofg), "frame %d" % t
    n = H * W
    w = eng.get_state("w", (5, n), np.float32)
    var = eng.get_state("var", (5, n), np.float32)
    nm = eng.get_state("nmodes", (n,), np.uint8)
    assert np.array_equal(w[:, idx], orc.get_state("w", (5, 65536), np.float32))
    assert np.array_equal(var[:, idx], orc.get_state("var", (5, 65536), np.float32))
    assert nm.min() >= 1 and nm.max() <= 5
    assert (np.diff(w, axis=0) <= 0).all(), "modes must stay sorted by weight"
    live = np.arange(5)[:, None] < nm[None, :]
    tot = np.where(live, w, 0).sum(0)  # == 1 after a renormalisation, < 1 right after a weakest-mode replacement
    assert (tot > 0.5).all() and (tot <= 1.0 + 1e-3).all()
    assert (var[live] >= 4.0).all() and (var[live] <= 75.0).all()


# ----------------------------------------------------------------------------- MOG1 / ASBL variants


def test_bench_geometry_32_streams_1080p_sampled_parity():
    """The bench.py workload itself (BASELINE configs[4] share of one GPU): 32 x 1920x1080 streams in one launch - a 6.7 GB model,
    i.e. byte offsets far past 2^32 - checked against the oracle on 2 048 random pixels of every stream (MOG2 is pointwise), plus
    the packed mask against the byte mask over all 66 M pixels."""
    torch = _torch()
    S, H, W, T = 32, 1080, 1920, 6
    eng = Engine(capi.MOG2, n_streams=S)
    eng.set_geometry(H, W, 3)
    rng = np.random.default_rng(2024)
    idx = np.sort(rng.choice(H * W, 2048, replace=False))
    idx[-1] = H * W - 1  # the very last pixel of every stream
    d_idx = torch.from_numpy(idx).cuda()
    orcs = [pyoracle.Oracle(capi.MOG2) for _ in range(S)]
    d_fg = torch.empty((S, H, W), dtype=torch.uint8, device="cuda")
    d_bits = torch.zeros((S, H * W // 64), dtype=torch.int64, device="cuda")
    for t in range(T):
        frames = torch.stack([synth.s_sat(1, H, W, seed=100 + s, device="cuda", t0=t)[0] for s in range(S)])
        eng.process_batch_device(frames, d_fg, None, d_bits)
        torch.cuda.synchronize()
        samp = frames.reshape(S, H * W, 3)[:, d_idx].cpu().numpy()
        got = d_fg.reshape(S, H * W)[:, d_idx].cpu().numpy()
        for s in range(S):
            ofg, _ = orcs[s].process(samp[s].reshape(32, 64, 3), want_bg=False)
            assert np.array_equal(got[s].reshape(32, 64), ofg), (t, s)
    bits = d_bits.cpu().numpy().view(np.uint8)
    unpacked = np.unpackbits(bits.reshape(S, -1), axis=1, bitorder="little")
    assert np.array_equal(unpacked != 0, d_fg.reshape(S, -1).cpu().numpy() != 0)
    for s in (0, S - 1):
        w = eng.get_state("w", (5, H * W), np.float32, stream=s)
        assert np.array_equal(w[:, idx], orcs[s].get_state("w", (5, 2048), np.float32)), s


def test_full_size_4k_wmv_and_abl_sampled_parity():
    """BASELINE configs[2]: WeightedMovingVarianceBGS + AdaptiveBackgroundLearning at 3840x2160, frames generated in HBM.
    Both are pointwise, so the oracle replays a 65 536-pixel random sample of the same 6-frame clip and must agree bit for bit;
    the whole frame is covered by a cross-check between the two device paths (byte mask vs bit-packed mask)."""
    torch = _torch()
    H, W, T = 2160, 3840, 6
    frames = synth.s_surv(T, H, W, seed=4321, device="cuda")
    rng = np.random.default_rng(7)
    idx = np.sort(rng.choice(H * W, 65536, replace=False))
    tidx = torch.from_numpy(idx).cuda()
    sample = frames.reshape(T, H * W, 3)[:, tidx].cpu().numpy().reshape(T, 256, 256, 3)
    for algo in (capi.WMV, capi.ABL):
        eng = Engine(algo)
        eng.set_geometry(H, W, 3)
        orc = pyoracle.Oracle(algo)
        d_fg = torch.empty((1, H, W), dtype=torch.uint8, device="cuda")
        d_bg = torch.empty((1, H, W, 3), dtype=torch.uint8, device="cuda")
        d_bits = torch.zeros((1, H * W // 64), dtype=torch.int64, device="cuda")
        for t in range(T):
            flags = eng.process_batch_device(frames[t:t + 1], d_fg, d_bg, d_bits)
            torch.cuda.synchronize()
            ofg, obg = orc.process(sample[t])
            assert bool(flags & capi.FG_VALID) == (ofg is not None)
            if ofg is not None:
                assert np.array_equal(d_fg.reshape(-1)[tidx].cpu().numpy().reshape(256, 256), ofg), (algo, t)
                bits = np.unpackbits(d_bits.cpu().numpy().view(np.uint8).reshape(-1), bitorder="little")
                assert np.array_equal(bits.astype(bool), d_fg.reshape(-1).cpu().numpy() != 0)
            if obg is not None:
                assert np.array_equal(d_bg.reshape(-1, 3)[tidx].cpu().numpy().reshape(256, 256, 3), obg), (algo, t)


def test_full_size_1080p_subsense_three_frames():
    """BASELINE configs[3]: SuBSENSE at 1920x1080 (5x5 diffusion, median 13, frame-level block on): three frames against the
    oracle (which needs ~6 s per frame at this size), mask + background + the learning-rate / threshold maps."""
    frames = synth.numpy_frames("surv", 3, 1080, 1920, seed=4321)
    eng, orc = Engine(capi.SUBSENSE), pyoracle.Oracle(capi.SUBSENSE)
    for f in frames:
        fg, bg = eng.process(f)
        ofg, obg = orc.process(f)
        assert np.array_equal(fg, ofg) and np.array_equal(bg, obg)
    n = 1080 * 1920
    for pl in ("R", "T", "V", "DminLT"):
        assert np.array_equal(eng.get_state(pl, (n,), np.float32), orc.get_state(pl, (n,), np.float32)), pl
    assert np.array_equal(eng.get_state("scalars", (7,), np.float64), orc.get_state("scalars", (7,), np.float64))


# ----------------------------------------------------------------------------- GMG


def test_lbsp_full_size_1080p():
    torch = _torch()
    from tracking_amd.engine import lbsp_describe_device
    img = synth.s_surv(1, 1080, 1920, seed=3, device="cuda")[0]
    lut = pyoracle.lbsp_lut(0.333, 0, 3)
    got = lbsp_describe_device(img, lut).cpu().numpy().view(np.uint16)
    want = pyoracle.lbsp_describe(img.cpu().numpy(), lut)
    assert np.array_equal(got, want)


# ----------------------------------------------------------------------------- mask post-processing primitives


def test_lbsp_matches_reference_fixture():
    """tests/golden/lbsp_ref.npz was produced by the reference's LBSP_16bits_dbcross_*.i (oracle/_ref) in the build container."""
    torch = _torch()
    from tracking_amd.engine import lbsp_describe_device
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "lbsp_ref.npz"))
    frames = np.load(os.path.join(os.path.dirname(__file__), "golden", "frames_96x80.npz"))["frames"]
    gray = np.load(os.path.join(os.path.dirname(__file__), "golden", "frames_gray_64x48.npz"))["frames"]
    d3 = lbsp_describe_device(torch.from_numpy(frames[0]).cuda(), g["lut3"]).cpu().numpy().view(np.uint16)
    assert np.array_equal(d3, g["desc3"])
    d7 = lbsp_describe_device(torch.from_numpy(frames[7]).cuda(), g["lut3"]).cpu().numpy().view(np.uint16)
    assert np.array_equal(d7, g["desc3_f7"])
    d1 = lbsp_describe_device(torch.from_numpy(gray[0]).cuda(), g["lut1"]).cpu().numpy().view(np.uint16)
    assert np.array_equal(d1[:, :, 0], g["desc1"])


@pytest.mark.parametrize("level", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("shape", [(64, 256), (37, 53)])
def test_mog2_sparse_levels_are_exact(level, shape):
    """BGS_OPT_MOG2_SPARSE only changes which planes move: masks, backgrounds and the whole model (including the stale entries
    past each pixel's mode count) must equal the oracle at every level.  The clip mixes quiet pixels (1 mode), a moving box
    (modes created / replaced) and a noisy band (all 5 modes live), so lanes and waves with different mode counts sit side by side."""
    torch = _torch()
    rng = np.random.default_rng(level * 10 + shape[0])
    H, W = shape
    T = 16
    base = rng.integers(0, 256, (H, W, 3))
    frames = np.repeat(base[None], T, 0).astype(np.int32)
    frames[:, :, : W // 5] = rng.integers(0, 256, (T, H, W // 5, 3))  # noisy band
    for t in range(T):
        x = (t * 5) % max(1, W - 12)
        frames[t, H // 3: H // 3 + 8, x: x + 12] = 255 - frames[t, H // 3: H // 3 + 8, x: x + 12]  # moving box
    frames = frames.astype(np.uint8)
    eng = Engine(capi.MOG2)
    eng.set_option(capi.OPT_MOG2_SPARSE, level)
    orc = pyoracle.Oracle(capi.MOG2)
    for t, f in enumerate(frames):
        fg, bg = eng.process(f)
        ofg, obg = orc.process(f)
        assert np.array_equal(fg, ofg) and np.array_equal(bg, obg), (level, t)
    check_mog2_state(eng, orc, H * W)


def _mog2_summary_invariants(eng, n, stream=0):
    """Whole-frame check of the summaries the filter kernel relies on (kernel_mog2.h): wherever a pixel's summaries are marked valid, each
    live slot's 16-bit summary - the 8-level bucket of each channel's mean (5 bits each) and a variance class - must cover its record:
    8 q_c - 2 <= mean_c <= 8 q_c + 9, and class 0 => var <= 32 - or the filter path could rule out a mode the reference would have
    matched.  Returns the fraction of pixels whose summaries are valid."""
    nm = eng.get_state("nmodes", (n,), np.uint8, stream=stream).astype(np.int32)
    valid = eng.get_state("summary_valid", (n,), np.uint8, stream=stream).astype(bool)
    mu = eng.get_state("mu", (5, 3, n), np.float32, stream=stream)
    var = eng.get_state("var", (5, n), np.float32, stream=stream)
    sm = eng.get_state("summary", (5, n), np.uint32, stream=stream)
    assert (sm >> 16).max() == 0, "16-bit words"
    live = (np.arange(5)[:, None] < nm[None, :]) & valid[None, :]
    cls = (sm >> 15).astype(bool)
    tight = live & ~cls  # class 1 promises nothing (such a mode is never ruled out)
    for c in range(3):
        b = (((sm >> (5 * c)) & 0x1f) << 3).astype(np.float32)
        e = mu[:, c] - b
        assert np.all((e >= -2.0)[tight]) and np.all((e <= 9.0)[tight]), ("mean", c, float(e[tight].min()), float(e[tight].max()))
    assert np.all((var <= 32.0)[tight]), "variance class"
    return float(valid.mean())


@pytest.mark.parametrize("level", [1, 2, 3, 4])
def test_mog2_summary_filter_boundary_and_invariants(level):
    """The record filter of MOG2's filter kernel (sparse 4; sparse 3 = auto runs it at least every 16th launch) against the oracle where it is most likely to slip: modes spaced 6 .. 34 grey levels apart (per column band), so
    that for neighbouring pixels the same comparison is ruled out by the summary, or barely not, or needs the record; noise and a slow
    drift keep the means moving across the summaries' rounding.  Masks, backgrounds every third frame (that launch reads every mode)
    and the whole model must equal the oracle; wherever a pixel's summaries are marked valid they must cover its records - after
    per-frame launches, after clip launches (which invalidate them) and after the per-frame launches that rebuild them."""
    torch = _torch()
    rng = np.random.default_rng(99 + level)
    H, W, T = 48, 320, 90
    spacing = np.repeat(np.array([6, 8, 10, 12, 14, 17, 20, 24, 28, 34]), W // 10)[None, :, None]
    base = rng.integers(10, 60, (H, W, 3))
    phase = rng.integers(0, 5, (H, W, 1))
    frames = np.empty((T, H, W, 3), np.uint8)
    for t in range(T):
        lvl = (t + phase) % (3 + (t // 30))  # 3, then 4, then 5 levels in play
        f = base + spacing * lvl + rng.integers(-3, 4, (H, W, 3)) + t // 12
        frames[t] = np.clip(f, 0, 255).astype(np.uint8)
    eng = Engine(capi.MOG2)
    eng.set_option(capi.OPT_MOG2_SPARSE, level)
    orc = pyoracle.Oracle(capi.MOG2)
    for t in range(60):
        want_bg = t % 3 == 0
        fg, bg = eng.process(frames[t], want_bg=want_bg)
        ofg, obg = orc.process(frames[t], want_bg=want_bg)
        assert np.array_equal(fg, ofg), (level, t, int((fg != ofg).sum()))
        if want_bg:
            assert np.array_equal(bg, obg), (level, t)
    check_mog2_state(eng, orc, H * W)
    vfrac = _mog2_summary_invariants(eng, H * W)
    # frame 59 delivered no background: level 4 ran the filter kernel; levels 1 / 2 never do; auto (3) runs whichever kernel it chose
    assert vfrac == 1.0 if level == 4 else vfrac == 0.0 if level < 3 else vfrac in (0.0, 1.0), (level, vfrac)
    # the rest of the clip through clip launches (8 + 8 + 8 + 4 + 2 frames): they change records without looking after the summaries
    dev = torch.from_numpy(frames[60:]).cuda().unsqueeze(1)
    fgd = torch.empty((30, 1, H, W), dtype=torch.uint8, device="cuda")
    eng.process_clip_device(dev, 30, fgd)
    torch.cuda.synchronize()
    for t in range(60, 90):
        ofg, _ = orc.process(frames[t], want_bg=False)
        assert np.array_equal(fgd[t - 60, 0].cpu().numpy(), ofg), (level, t)
    check_mog2_state(eng, orc, H * W)
    assert _mog2_summary_invariants(eng, H * W) == 0.0
    for t in range(5):  # per-frame again: the first launch rebuilds the summaries, the next ones use them
        fg, _ = eng.process(frames[t], want_bg=False)
        ofg, _ = orc.process(frames[t], want_bg=False)
        assert np.array_equal(fg, ofg), (level, "after clip", t)
        vfrac = _mog2_summary_invariants(eng, H * W)
        assert vfrac == 1.0 if level == 4 else vfrac == 0.0 if level < 3 else vfrac in (0.0, 1.0), (level, t, vfrac)
    check_mog2_state(eng, orc, H * W)
    eng.close()


def test_the_state_bench_times_aged_1080p_s_sat_model_filter_kernel_steady():
    """What bench.py's timed region runs on: 1080p S_sat streams whose model is OLDER than 100 frames (the five weights of a pixel have
    equalised, every frame re-orders the modes) with auto mode settled on the filter kernel - the earlier 1080p tests stop at 20 / 6
    frames, where count and filter launches still alternate.  2 streams x 140 frames with fresh noise in every frame (the bench's
    own source, tools/synth.py SatStreams); the oracle replays 4 096 sampled pixels of each stream: every mask bit-exact, the sampled
    model state equal, and over ALL pixels the summaries' invariants, sorted weights, variance clamps."""
    torch = _torch()
    S, H, W, T, NS = 2, 1080, 1920, 140, 4096
    eng = Engine(capi.MOG2, n_streams=S)
    eng.set_geometry(H, W, 3)
    src = synth.SatStreams(S, H, W, seed0=1234, device="cuda")
    rng = np.random.default_rng(31)
    idx = np.sort(rng.choice(H * W, NS, replace=False))
    idx[0], idx[-1] = 0, H * W - 1
    d_idx = torch.from_numpy(idx).cuda()
    orcs = [pyoracle.Oracle(capi.MOG2) for _ in range(S)]
    cur = torch.empty((S, H, W, 3), dtype=torch.uint8, device="cuda")
    d_fg = torch.empty((S, H, W), dtype=torch.uint8, device="cuda")
    samp = torch.empty((T, S, NS, 3), dtype=torch.uint8, device="cuda")
    got = torch.empty((T, S, NS), dtype=torch.uint8, device="cuda")
    eng.enable_kernel_timing(True)
    for t in range(T):
        src.into(cur)
        eng.process_batch_device(cur, d_fg, None, None)
        samp[t] = cur.reshape(S, H * W, 3)[:, d_idx]
        got[t] = d_fg.reshape(S, H * W)[:, d_idx]
    torch.cuda.synchronize()
    samp, got = samp.cpu().numpy(), got.cpu().numpy()
    for s in range(S):
        for t in range(T):
            ofg, _ = orcs[s].process(samp[t, s].reshape(64, 64, 3), want_bg=False)
            assert np.array_equal(got[t, s].reshape(64, 64), ofg), (s, t, int((got[t, s].reshape(64, 64) != ofg).sum()))
    n = H * W
    for s in range(S):
        w = eng.get_state("w", (5, n), np.float32, stream=s)
        var = eng.get_state("var", (5, n), np.float32, stream=s)
        mu = eng.get_state("mu", (5, 3, n), np.float32, stream=s)
        nm = eng.get_state("nmodes", (n,), np.uint8, stream=s)
        for name, a, shape in (("w", w[:, idx], (5, NS)), ("var", var[:, idx], (5, NS)), ("mu", mu[:, :, idx], (5, 3, NS))):
            err = max_err(a, orcs[s].get_state(name, shape, np.float32))
            assert err <= STATE_TOL, (s, name, err)
        assert np.array_equal(nm[idx], orcs[s].get_state("nmodes", (NS,), np.uint8))
        assert (nm == 5).mean() > 0.999, "S_sat: all five modes live"
        assert (np.diff(w, axis=0) <= 0).all(), "modes stay sorted by weight"
        assert (var >= 4.0).all() and (var <= 75.0).all()
        # the aged model's steady state: auto mode has settled on the filter kernel, so every pixel's summaries are valid and cover its records
        assert _mog2_summary_invariants(eng, n, stream=s) == 1.0
    series = eng.kernel_timing_series()
    assert len(series) == T
    eng.close()


@pytest.mark.parametrize("algo,S,T", [(capi.SUBSENSE, 8, 3), (capi.MOG1, 16, 4), (capi.DP_GRIMSON_GMM, 32, 3)])
def test_large_batches_match_single_stream_engines(algo, S, T):
    """Models past 4 GB (SuBSENSE: 8 x 1080p x 50 samples = 7.5 GB; MOG1: 16 x 1080p = 5.3 GB; Grimson: 32 x 1080p = 4.8 GB): the
    last stream of the batch must equal a single-stream engine fed the same frames (which the other tests hold against the
    oracle) - the 64-bit offset check for the layouts the oracle is too slow to replay at this size."""
    torch = _torch()
    H, W = 1080, 1920
    clips = [synth.s_surv(T, H, W, seed=900 + s, device="cuda") for s in (0, S - 1)]
    filler = synth.s_surv(1, H, W, seed=77, device="cuda")[0]
    big = Engine(algo, n_streams=S)
    big.set_geometry(H, W, 3)
    singles = [Engine(algo), Engine(algo)]
    for e in singles:
        e.set_geometry(H, W, 3)
    fg = torch.empty((S, H, W), dtype=torch.uint8, device="cuda")
    fg1 = torch.empty((1, H, W), dtype=torch.uint8, device="cuda")
    frames = filler.unsqueeze(0).repeat(S, 1, 1, 1)
    for t in range(T):
        frames[0], frames[S - 1] = clips[0][t], clips[1][t]
        big.process_batch_device(frames, fg, None, None)
        for k, s in enumerate((0, S - 1)):
            singles[k].process_batch_device(clips[k][t:t + 1], fg1, None, None)
            torch.cuda.synchronize()
            assert torch.equal(fg[s], fg1[0]), (t, s)


def test_lobster_golden_frames_and_scene_change(golden_frames):
    """LOBSTERBGS (N4): masks, backgrounds and the whole sample model (35 colour + descriptor samples) equal the oracle under the
    two-phase / counter-RNG contract it shares with SuBSENSE; the second half of the clip is brightness-shifted."""
    shifted = np.clip(golden_frames.astype(np.int32) + 50, 0, 255).astype(np.uint8)
    frames = np.concatenate([golden_frames, shifted[:12]])
    eng, orc, outs = run_pair(capi.LOBSTER, frames)
    check_lobster_state(eng, orc, frames.shape[1], frames.shape[2])
    assert outs[0][0].max() == 0 and outs[len(golden_frames)][0].mean() > 50  # first frame: all background; after the cut: mostly foreground


@pytest.mark.parametrize("name", DP_NAMES)
def test_dp_models_long_clip_with_scene_changes(name, golden_frames):
    """package_bgs/dp (N4): 72 frames = the golden clip, a brightness-shifted copy and its reverse, so that modes are created,
    matched, re-sorted, pruned and replaced (GMMs), the median walks, and the single gaussian saturates its variance clamp."""
    shifted = np.clip(golden_frames.astype(np.int32) + 60, 0, 255).astype(np.uint8)
    frames = np.concatenate([golden_frames, shifted, golden_frames[::-1]])
    eng, orc, _ = run_pair(ALGOS[name], frames)
    check_dp_state(name, eng, orc, frames.shape[1] * frames.shape[2])


@pytest.mark.parametrize("algo", [capi.SUBSENSE, capi.LOBSTER])
def test_sample_consensus_models_packed_mask(algo):
    """bgs_process_batch_device with d_bits for SuBSENSE / LOBSTER: the bit mask MaskGather ships equals the byte mask."""
    torch = _torch()
    S, T, H, W = 2, 5, 48, 64
    eng = Engine(algo, n_streams=S)
    eng.set_geometry(H, W, 3)
    clips = np.stack([synth.random_frames(T, H, W, 3, seed=300 + s) for s in range(S)])
    clips[:, 2:] = clips[:, :1]  # repeat the first frame so that part of the mask goes to background
    for t in range(T):
        d_frames = torch.from_numpy(np.ascontiguousarray(clips[:, t])).cuda()
        d_fg = torch.empty((S, H, W), dtype=torch.uint8, device="cuda")
        d_bits = torch.zeros((S, H * W // 64), dtype=torch.int64, device="cuda")
        eng.process_batch_device(d_frames, d_fg, None, d_bits)
        torch.cuda.synchronize()
        bits = np.unpackbits(d_bits.cpu().numpy().view(np.uint8).reshape(S, -1), axis=1, bitorder="little").reshape(S, H, W)
        assert np.array_equal(bits != 0, d_fg.cpu().numpy() != 0), t
