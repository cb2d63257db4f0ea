"""Independent per-camera lifetime inside one batched engine (one IBGS object per stream, created and deleted whenever the
caller likes - FrameProcessor.cpp:35-155, :342-482; ustc_src/ustc_bgs.cpp:75-77): streams of one engine join at different frames, skip
frames, are reset mid-run, and every one of them must equal ITS OWN oracle instance at every frame - through the per-frame device
path (one call over streams of different ages) and through clip calls."""
import numpy as np
import pytest

from gpu_helpers import ALGOS, _torch, check_lobster_state, check_state, check_subsense_state
from oracle import pyoracle
from tools import synth
from tracking_amd import Engine, capi

pytestmark = pytest.mark.gpu

NAMES = sorted(ALGOS) + ["SuBSENSEBGS", "LOBSTERBGS"]
ALL = dict(ALGOS, SuBSENSEBGS=capi.SUBSENSE, LOBSTERBGS=capi.LOBSTER)


def _compare(name, algo, eng, orcs, s, frame, d_fg, d_bg, H, W, tag):
    ofg, obg = orcs[s].process(frame)
    fl = eng.stream_flags(s)
    assert bool(fl & capi.FG_VALID) == (ofg is not None), (name, tag, s, fl)
    assert bool(fl & capi.BG_VALID) == (obg is not None), (name, tag, s, fl)
    if ofg is not None:
        got = d_fg.cpu().numpy()
        assert np.array_equal(got, ofg), (name, tag, s, int((got != ofg).sum()))
    else:
        assert bool((d_fg == 9).all()), (name, tag, s, "a warm-up frame must leave the output untouched")
    if obg is not None:
        assert np.array_equal(d_bg.cpu().numpy().reshape(obg.shape), obg), (name, tag, s)


@pytest.mark.parametrize("name", NAMES)
def test_streams_join_skip_and_reset_independently(name):
    torch = _torch()
    algo = ALL[name]
    S, T, H, W = 4, 14, 32, 64
    clips = np.stack([synth.random_frames(T, H, W, 3, seed=500 + s) for s in range(S)])  # [S][T][H][W][3]
    eng = Engine(algo, n_streams=S)
    eng.set_geometry(H, W, 3)
    orcs = [pyoracle.Oracle(algo) for _ in range(S)]
    fed = [0] * S                     # frames each stream has been given so far (its own clock)
    seen = [0] * S                    # ... since its last reset
    join = [0, 2, 3, 5]               # step at which a camera comes up
    skip = {(2, 7), (0, 9)}           # (stream, step): this camera drops this step's frame
    reset_at = {8: 1, 11: 3}          # step -> stream whose IBGS object is deleted and created again
    bg_c = 1 if algo == capi.ASBL else 3
    for step in range(T):
        if step in reset_at:
            r = reset_at[step]
            eng.reset_stream(r)
            orcs[r].close()
            orcs[r] = pyoracle.Oracle(algo)
            seen[r] = 0
            assert eng.frames_seen(r) == 0
        active = [s for s in range(S) if join[s] <= step and (s, step) not in skip]
        # contiguous groups of active streams go through ONE call each, whatever their ages
        groups, cur = [], []
        for s in range(S):
            if s in active:
                cur.append(s)
            elif cur:
                groups.append(cur)
                cur = []
        if cur:
            groups.append(cur)
        for g in groups:
            frames = np.stack([clips[s, fed[s]] for s in g])
            d_frames = torch.from_numpy(frames).cuda()
            d_fg = torch.full((len(g), H, W), 9, dtype=torch.uint8, device="cuda")
            d_bg = torch.full((len(g), H, W, bg_c), 9, dtype=torch.uint8, device="cuda")
            eng.process_batch_device(d_frames, d_fg, d_bg, None, first=g[0], count=len(g))
            torch.cuda.synchronize()
            for k, s in enumerate(g):
                _compare(name, algo, eng, orcs, s, clips[s, fed[s]], d_fg[k], d_bg[k], H, W, ("step", step))
                fed[s] += 1
                seen[s] += 1
                assert eng.frames_seen(s) == seen[s]
    for s in range(S):
        if name == "SuBSENSEBGS":
            check_subsense_state(eng, orcs[s], H, W, stream=s)
        elif name == "LOBSTERBGS":
            check_lobster_state(eng, orcs[s], H, W, stream=s)
        else:
            check_state(name, eng, orcs[s], H * W, stream=s)
    eng.close()
    for o in orcs:
        o.close()


@pytest.mark.parametrize("name", ["MixtureOfGaussianV2BGS", "MixtureOfGaussianV1BGS", "DPZivkovicAGMMBGS", "FrameDifferenceBGS", "WeightedMovingVarianceBGS", "AdaptiveBackgroundLearning"])
def test_clip_call_over_streams_of_different_ages(name):
    """bgs_process_clip_device over a range whose streams have seen 0, 3, 3 and 1 frames: each stream's masks over the clip equal its
    own oracle's; then one more per-frame call over all four (ages 6, 9, 9, 7)."""
    torch = _torch()
    algo = ALL[name]
    S, H, W, NC = 4, 32, 64, 6
    ages = [0, 3, 3, 1]
    clips = np.stack([synth.random_frames(12, H, W, 3, seed=900 + s) for s in range(S)])
    eng = Engine(algo, n_streams=S)
    eng.set_geometry(H, W, 3)
    orcs = [pyoracle.Oracle(algo) for _ in range(S)]
    for s in range(S):  # bring every stream to its age with single-stream calls
        for t in range(ages[s]):
            d = torch.from_numpy(np.ascontiguousarray(clips[s, t])).cuda().unsqueeze(0)
            eng.process_batch_device(d, None, None, None, first=s, count=1)
            orcs[s].process(clips[s, t])
    slab = np.stack([np.stack([clips[s, ages[s] + t] for s in range(S)]) for t in range(NC)])  # [NC][S][H][W][3]
    d_slab = torch.from_numpy(slab).cuda()
    d_fg = torch.full((NC, S, H, W), 9, dtype=torch.uint8, device="cuda")
    eng.process_clip_device(d_slab, NC, d_fg)
    torch.cuda.synchronize()
    fg = d_fg.cpu().numpy()
    for s in range(S):
        for t in range(NC):
            ofg, _ = orcs[s].process(clips[s, ages[s] + t], want_bg=False)
            if ofg is not None:
                assert np.array_equal(fg[t, s], ofg), (name, s, t)
            else:
                assert (fg[t, s] == 9).all(), (name, s, t)
        assert eng.frames_seen(s) == ages[s] + NC
    frames = np.stack([clips[s, ages[s] + NC] for s in range(S)])
    d_fg1 = torch.full((S, H, W), 9, dtype=torch.uint8, device="cuda")
    eng.process_batch_device(torch.from_numpy(frames).cuda(), d_fg1, None, None)
    torch.cuda.synchronize()
    for s in range(S):
        ofg, _ = orcs[s].process(frames[s], want_bg=False)
        assert np.array_equal(d_fg1[s].cpu().numpy(), ofg), (name, s, "after clip")
        check_state(name, eng, orcs[s], H * W, stream=s)
    eng.close()


def test_chunked_model_is_released_on_destroy():
    """A model built from separately created physical chunks (hipMemCreate / hipMemMap) is given back on bgs_destroy:
    create / run / destroy cycles of an engine whose model takes that construction (threshold and chunk size lowered so that a
    25 MB model is 7 chunks) leave the device's free memory where it was; results equal the plainly allocated engine's bit for bit."""
    torch = _torch()
    S, H, W = 4, 135, 384   # 207 360 pixels = 810 tiles of 256: 25.3 MB of model
    frames = torch.from_numpy(np.stack([synth.random_frames(3, H, W, 3, seed=40 + s) for s in range(S)])).cuda()  # [S][3][H][W][3]
    ref = None
    free0 = None
    fg = torch.empty((S, H, W), dtype=torch.uint8, device="cuda")
    for cycle in range(5):
        if cycle == 2:  # baseline after one plain and one chunked engine have come and gone (runtime pools, code objects, streams are warm)
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info()[0]
        eng = Engine(capi.MOG2, n_streams=S)
        if cycle:  # cycle 0: one plain hipMalloc (the reference result)
            eng.set_option(capi.OPT_MODEL_CHUNK_MIN_MB, 1)
            eng.set_option(capi.OPT_MODEL_CHUNK_MB, 4)
        eng.set_geometry(H, W, 3)
        pr = eng.get_state("placement", (2,), np.float32)
        assert (int(pr[1]) >= 6) == bool(cycle), pr
        for t in range(3):
            eng.process_batch_device(frames[:, t].contiguous(), fg, None, None)
        torch.cuda.synchronize()
        w = eng.get_state("w", (5, H * W), np.float32, stream=S - 1)
        if ref is None:
            ref = (fg.cpu().numpy().copy(), w)
        else:
            assert np.array_equal(fg.cpu().numpy(), ref[0]) and np.array_equal(w, ref[1])
        eng.close()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    # three chunked engines of 25 MB each were created and destroyed after the baseline: a leaked model shows as >= 25 MB
    assert free0 - free1 < (12 << 20), "device memory not returned: %d bytes missing after 3 chunked engines" % (free0 - free1)


def test_device_calls_refuse_a_stream_with_a_submission_in_flight():
    """bgs_submit runs on its lane's own HIP stream: a device-path call that covers that camera before bgs_wait would race with it and is
    refused (BGS_ERR_STATE); other cameras are served, and after bgs_wait the call goes through."""
    torch = _torch()
    S, H, W = 3, 48, 64
    clip = synth.random_frames(4, H, W, 3, seed=77)
    eng = Engine(capi.MOG2, n_streams=S)
    fgs = [np.empty((H, W), np.uint8) for _ in range(S)]
    for s in range(S):
        eng.submit(np.ascontiguousarray(clip[0]), fgs[s], None, stream=s)
    for s in range(S):
        eng.wait(stream=s)
    eng.submit(np.ascontiguousarray(clip[1]), fgs[1], None, stream=1)
    d = torch.from_numpy(np.stack([clip[1]] * S)).cuda()
    d_fg = torch.empty((S, H, W), dtype=torch.uint8, device="cuda")
    with pytest.raises(capi.BgsError) as ei:
        eng.process_batch_device(d, d_fg, None, None)
    assert "in flight" in str(ei.value)
    with pytest.raises(capi.BgsError):
        eng.process_clip_device(d.unsqueeze(0), 1, d_fg.unsqueeze(0))
    eng.process_batch_device(d[:1], d_fg[:1], None, None, first=0, count=1)  # camera 0 has nothing in flight
    eng.wait(stream=1)
    eng.process_batch_device(d[1:], d_fg[1:], None, None, first=1, count=2)
    torch.cuda.synchronize()
    assert [eng.frames_seen(s) for s in range(S)] == [2, 3, 2]
    eng.close()


def test_host_arena_cameras_dma_in_place_equal_the_oracle():
    """bgs_host_arena: all cameras' frames in ONE array and all their masks / backgrounds in others, each page-locked once; every image
    inside an arena is read / written by the DMA engine in place (no staging copy: the engine's own counter says so) through
    bgs_submit / bgs_wait and through bgs_process; results equal every camera's own oracle; dropping the arena falls back to staging."""
    _torch()
    S, T, H, W = 3, 6, 96, 160
    clips = np.stack([synth.random_frames(T, H, W, 3, seed=210 + s) for s in range(S)])
    eng = Engine(capi.MOG2, n_streams=S)
    def own_pages(shape):  # an arena on pages of its own (small numpy arrays share heap pages; a page cannot be page-locked twice)
        n = int(np.prod(shape))
        raw = np.empty(n + 2 * 4096, np.uint8)
        off = (-raw.ctypes.data) % 4096
        return raw[off:off + (n + 4095) // 4096 * 4096][:n].reshape(shape), raw
    (frames, k0), (fgs, k1), (bgs_, k2) = own_pages((S, H, W, 3)), own_pages((S, H, W)), own_pages((S, H, W, 3))
    eng.set_geometry(H, W, 3)
    for a in (frames, fgs, bgs_):
        eng.host_arena(a)
    orcs = [pyoracle.Oracle(capi.MOG2) for _ in range(S)]
    for t in range(T):
        frames[...] = clips[:, t]
        if t % 2 == 0:
            for s in range(S):
                eng.submit(frames[s], fgs[s], bgs_[s], stream=s)
            for s in range(S):
                eng.wait(stream=s)
        else:
            for s in range(S):
                eng.process_into(frames[s], fgs[s], bgs_[s], stream=s)
        for s in range(S):
            ofg, obg = orcs[s].process(clips[s, t])
            assert np.array_equal(fgs[s], ofg) and np.array_equal(bgs_[s], obg), (t, s)
    d = eng.get_state("hostpath", (15,), np.float64)
    assert d[14] == 3 and d[6] == 3 and d[12] == 0 and d[13] == 0, d  # three arenas, three registrations, no CPU staging copies either way
    eng.host_arena(frames, on=False)
    frames[...] = clips[:, 0]
    eng.process_into(frames[0], fgs[0], bgs_[0], stream=0)
    d2 = eng.get_state("hostpath", (15,), np.float64)
    assert d2[14] == 2 and d2[12] > 0, d2  # the input goes through the pinned staging buffer again
    ofg, _ = orcs[0].process(clips[0, 0])
    assert np.array_equal(fgs[0], ofg)
    eng.close()
