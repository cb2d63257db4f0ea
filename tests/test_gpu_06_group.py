"""bgs_group - several classes on the same frames (what FrameProcessor::process does with its pre-processed frame,
FrameProcessor.cpp:169-340): every output of a group must equal the oracle of its class and the separate engine of its class, bit for
bit - masks, backgrounds, warm-up conventions (outputs untouched), model states - whether the class runs inside the fused kernel or
as a member engine, on the host path and on the batched device path."""
import numpy as np
import pytest

from gpu_helpers import _params, _torch, check_mog2_state
from oracle import pyoracle
from tools import synth
from tracking_amd import Engine, capi
from tracking_amd.engine import Group

pytestmark = pytest.mark.gpu

FUSABLE = [capi.FRAME_DIFF, capi.STATIC_FRAME_DIFF, capi.WMM, capi.WMV, capi.ABL, capi.SIGMA_DELTA]


def _check_fused_state(grp, i, algo, orc, n):
    if algo in (capi.STATIC_FRAME_DIFF, capi.ABL):
        assert np.array_equal(grp.get_state(i, "bg", (n * 3,), np.uint8), orc.get_state("bg", (n * 3,), np.uint8)), (algo, "bg")
    if algo == capi.SIGMA_DELTA:
        for pl in ("mt", "vt"):
            assert np.array_equal(grp.get_state(i, pl, (n * 3,), np.uint8), orc.get_state(pl, (n * 3,), np.uint8)), pl


@pytest.mark.parametrize("algos", [FUSABLE, [capi.WMV, capi.ABL], [capi.FRAME_DIFF, capi.MOG2, capi.ABL, capi.ASBL],
                                   [capi.ABL, capi.ABL, capi.SIGMA_DELTA], [capi.WMM]])
@pytest.mark.parametrize("shape", [(48, 64), (37, 53)])
def test_group_host_path_equals_every_class_oracle(algos, shape, golden_frames):
    H, W = shape
    frames = np.ascontiguousarray(golden_frames[:14, :H, :W])
    grp = Group(algos)
    orcs = [pyoracle.Oracle(a) for a in algos]
    seen_first = set()
    for i, a in enumerate(algos):  # one instance of each byte-stream class is fused, everything else is a member engine
        assert grp.is_fused(i) == (a in FUSABLE and a not in seen_first), (i, a)
        seen_first.add(a)
    for t, f in enumerate(frames):
        outs = grp.process(f)
        for i, (a, orc) in enumerate(zip(algos, orcs)):
            ofg, obg = orc.process(f)
            fg, bg = outs[i]
            assert (fg is None) == (ofg is None), (t, i, a)
            assert (bg is None) == (obg is None), (t, i, a)
            if ofg is not None:
                assert np.array_equal(fg, ofg), (t, i, a, int((fg != ofg).sum()))
            if obg is not None:
                assert np.array_equal(bg.reshape(obg.shape), obg), (t, i, a)
    for i, (a, orc) in enumerate(zip(algos, orcs)):
        if grp.is_fused(i):
            _check_fused_state(grp, i, a, orc, H * W)
    assert grp.frames_seen() == len(frames)
    grp.close()


def test_group_parameter_changes_between_frames(golden_frames):
    """The wrappers re-read ./config/<Class>.xml on every process(): thresholds, weights and ABL's alpha (its 256 x 256 table is rebuilt)
    may change mid-stream, per class."""
    frames = golden_frames[:12]
    algos = [capi.WMV, capi.ABL, capi.FRAME_DIFF]
    grp = Group(algos)
    orcs = [pyoracle.Oracle(a) for a in algos]
    for t, f in enumerate(frames):
        if t == 5:
            pa, pw, pf = _params(capi.ABL, alpha=0.2, threshold=9), _params(capi.WMV, enable_weight=0, threshold=21), _params(capi.FRAME_DIFF, enable_threshold=0)
            for i, p in ((0, pw), (1, pa), (2, pf)):
                grp.set_params(i, p)
                orcs[i].set_params(p)
        outs = grp.process(f)
        for i, orc in enumerate(orcs):
            ofg, obg = orc.process(f)
            if ofg is not None:
                assert np.array_equal(outs[i][0], ofg), (t, i)
            else:
                assert outs[i][0] is None
            if obg is not None:
                assert np.array_equal(outs[i][1].reshape(obg.shape), obg), (t, i)
    grp.close()


@pytest.mark.parametrize("borrow", [False, True])
def test_group_device_batch_equals_separate_engines_and_oracles(borrow):
    """Batched device path, 3 streams: the fused kernel's outputs against separate engines of each class (same device frames) and
    against one oracle per stream and class; warm-up frames leave the outputs untouched."""
    torch = _torch()
    S, T, H, W = 3, 9, 32, 64
    algos = [capi.FRAME_DIFF, capi.STATIC_FRAME_DIFF, capi.WMM, capi.WMV, capi.ABL, capi.SIGMA_DELTA, capi.MOG2]
    clips = np.stack([synth.random_frames(T, H, W, 3, seed=300 + s) for s in range(S)])
    grp = Group(algos, n_streams=S)
    grp.set_geometry(H, W, 3)
    if borrow:
        grp.set_option(capi.OPT_BORROW_FRAMES, 1)
    engs = [Engine(a, n_streams=S) for a in algos]
    for e in engs:
        e.set_geometry(H, W, 3)
    orcs = [[pyoracle.Oracle(a) for _ in range(S)] for a in algos]
    keep = []
    for t in range(T):
        d_frames = torch.from_numpy(np.ascontiguousarray(clips[:, t])).cuda()
        keep.append(d_frames)  # borrowed history must stay alive
        fgs = [torch.full((S, H, W), 9, dtype=torch.uint8, device="cuda") for _ in algos]
        bgs_ = [torch.full((S, H, W, 3), 9, dtype=torch.uint8, device="cuda") for _ in algos]
        flags = grp.process_batch_device(d_frames, fgs, bgs_)
        torch.cuda.synchronize()
        for i, a in enumerate(algos):
            efg = torch.full((S, H, W), 9, dtype=torch.uint8, device="cuda")
            ebg = torch.full((S, H, W, 3), 9, dtype=torch.uint8, device="cuda")
            efl = engs[i].process_batch_device(d_frames, efg, ebg, None)
            torch.cuda.synchronize()
            assert flags[i] == efl, (t, i, a, flags[i], efl)
            assert torch.equal(fgs[i], efg), (t, i, a, "mask vs separate engine")
            assert torch.equal(bgs_[i], ebg), (t, i, a, "background vs separate engine")
            for s in range(S):
                ofg, obg = orcs[i][s].process(clips[s, t])
                assert bool(flags[i] & capi.FG_VALID) == (ofg is not None) and bool(flags[i] & capi.BG_VALID) == (obg is not None), (t, i, s)
                if ofg is not None:
                    assert np.array_equal(fgs[i][s].cpu().numpy(), ofg), (t, i, a, s)
                if obg is not None:
                    assert np.array_equal(bgs_[i][s].cpu().numpy().reshape(obg.shape), obg), (t, i, a, s)
    for s in range(S):
        check_mog2_state(engs[6], orcs[6][s], H * W, stream=s)
        mog2 = [grp.get_state(6, pl, sh, np.float32, stream=s) for pl, sh in (("w", (5, H * W)), ("var", (5, H * W)), ("mu", (5, 3, H * W)))]
        ref = [engs[6].get_state(pl, sh, np.float32, stream=s) for pl, sh in (("w", (5, H * W)), ("var", (5, H * W)), ("mu", (5, 3, H * W)))]
        assert all(np.array_equal(a, b) for a, b in zip(mog2, ref)), "member engine state"
    grp.close()


def test_group_config2_full_size_wmv_abl_sampled_parity():
    """BASELINE configs[2] as one fused launch: WeightedMovingVarianceBGS + AdaptiveBackgroundLearning on the same 3840x2160 frames,
    borrowed history (17 B/pixel).  Every pixel against the two separate engines; a band of rows of each against the oracles."""
    torch = _torch()
    H, W, T = 2160, 3840, 5
    frames = synth.s_surv(T, H, W, seed=77, device="cuda")
    grp = Group([capi.WMV, capi.ABL])
    grp.set_geometry(H, W, 3)
    grp.set_option(capi.OPT_BORROW_FRAMES, 1)
    ew, ea = Engine(capi.WMV), Engine(capi.ABL)
    for e in (ew, ea):
        e.set_geometry(H, W, 3)
    ew.set_option(capi.OPT_BORROW_FRAMES, 1)
    y0, y1 = 1000, 1064
    ow, oa = pyoracle.Oracle(capi.WMV), pyoracle.Oracle(capi.ABL)
    for t in range(T):
        fr = frames[t:t + 1]
        fgs = [torch.full((1, H, W), 9, dtype=torch.uint8, device="cuda") for _ in range(2)]
        bg = torch.empty((1, H, W, 3), dtype=torch.uint8, device="cuda")
        fl = grp.process_batch_device(fr, fgs, [None, bg])
        f1 = torch.full((1, H, W), 9, dtype=torch.uint8, device="cuda")
        f2 = torch.full((1, H, W), 9, dtype=torch.uint8, device="cuda")
        b2 = torch.empty((1, H, W, 3), dtype=torch.uint8, device="cuda")
        assert fl[0] == ew.process_batch_device(fr, f1, None, None) and fl[1] == ea.process_batch_device(fr, f2, b2, None)
        torch.cuda.synchronize()
        assert torch.equal(fgs[0], f1) and torch.equal(fgs[1], f2) and torch.equal(bg, b2), t
        band = np.ascontiguousarray(frames[t, y0:y1].cpu().numpy())
        wfg, _ = ow.process(band)
        afg, abg = oa.process(band)
        if wfg is not None:
            assert np.array_equal(fgs[0][0, y0:y1].cpu().numpy(), wfg), t
        assert np.array_equal(fgs[1][0, y0:y1].cpu().numpy(), afg) and np.array_equal(bg[0, y0:y1].cpu().numpy(), abg), t
    grp.close()


def test_group_argument_errors():
    with pytest.raises(capi.BgsError):
        Group([capi.SIGMA_DELTA, capi.ABL]).process(np.zeros((8, 8), np.uint8))  # SigmaDelta is 3-channel only
    g = Group([capi.WMV, capi.ABL], n_streams=2)
    with pytest.raises(capi.BgsError):
        g.process(np.zeros((8, 8, 3), np.uint8))  # the host call serves single-stream groups
    g.close()
    g = Group([capi.FRAME_DIFF, capi.WMV])
    g.process(np.zeros((8, 16, 3), np.uint8))
    with pytest.raises(capi.BgsError):
        g.process(np.zeros((9, 16, 3), np.uint8))  # a group keeps its geometry
    g.close()
