"""N3: the frame preparation in front of the path - VideoCapture's resize / flip / ROI and PreProcessor's
equalizeHist / GaussianBlur - on the device, against the CPU restatement (oracle/ingest_oracle.c).  Bit-exact (byte work).
Flip and ROI are exact by definition; the OpenCV arithmetic of the other three is recalled: parity unpinned."""
import numpy as np
import pytest

from oracle import pyoracle
from tools import synth
from tracking_amd import Engine, capi
from tracking_amd.engine import ingest_device, ingest_host

pytestmark = pytest.mark.gpu

CONFIGS = [
    dict(),
    dict(flip=1),
    dict(roi_x0=3, roi_y0=5, roi_x1=41, roi_y1=30),
    dict(flip=1, roi_x0=8, roi_y0=0, roi_x1=48, roi_y1=33),
    dict(resize_percent=50),
    dict(resize_percent=75, flip=1),
    dict(resize_percent=130, roi_x0=10, roi_y0=10, roi_x1=60, roi_y1=40),
    dict(resize_percent=33),
    dict(gaussian_blur=1),
    dict(resize_percent=50, flip=1, gaussian_blur=1),
    dict(resize_percent=200, gaussian_blur=1, roi_x0=1, roi_y0=2, roi_x1=100, roi_y1=71),
]


@pytest.mark.parametrize("kw", CONFIGS)
@pytest.mark.parametrize("shape", [(48, 64), (37, 53)])
def test_ingest_device_vs_oracle_bgr(kw, shape):
    import torch
    frames = synth.random_frames(3, shape[0], shape[1], 3, seed=shape[0] + len(kw))
    cfg = capi.default_ingest(**kw)
    want = [pyoracle.ingest(cfg, f) for f in frames]
    if want[0] is None:  # ROI outside the (odd-sized) frame: the ABI must refuse it too
        with pytest.raises(Exception):
            ingest_device(cfg, torch.from_numpy(frames).cuda())
        return
    got = ingest_device(cfg, torch.from_numpy(frames).cuda()).cpu().numpy()
    for k in range(3):
        assert np.array_equal(got[k], want[k]), (kw, k, int((got[k] != want[k]).sum()))
    # host-buffer form, strided source (a ROI view of a wider image, like cv::Mat(frame) after cvSetImageROI)
    wide = np.zeros((shape[0], shape[1] + 11, 3), np.uint8)
    wide[:, 4:4 + shape[1]] = frames[0]
    assert np.array_equal(ingest_host(cfg, wide[:, 4:4 + shape[1]]), want[0])


@pytest.mark.parametrize("kw", [dict(equalize_hist=1), dict(equalize_hist=1, gaussian_blur=1), dict(resize_percent=50, equalize_hist=1, flip=1),
                                dict(gaussian_blur=1), dict(resize_percent=150)])
@pytest.mark.parametrize("shape", [(48, 64), (5, 7), (1, 1), (130, 70)])
def test_ingest_device_vs_oracle_gray(kw, shape):
    import torch
    rng = np.random.default_rng(shape[0])
    frames = rng.integers(20, 180, (2,) + shape, dtype=np.uint8)
    frames[1] = 99  # a constant image: equalizeHist's "all pixels in one bin" branch
    cfg = capi.default_ingest(**kw)
    want = [pyoracle.ingest(cfg, f) for f in frames]
    if want[0] is None:
        with pytest.raises(Exception):
            ingest_device(cfg, torch.from_numpy(frames).cuda())
        return
    got = ingest_device(cfg, torch.from_numpy(frames).cuda()).cpu().numpy()
    for k in range(2):
        assert np.array_equal(got[k], want[k]), (kw, k)


def test_ingest_rejects_what_the_reference_fails_on():
    import torch
    f = torch.zeros((1, 20, 20, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(Exception):
        ingest_device(capi.default_ingest(equalize_hist=1), f)  # cv::equalizeHist asserts CV_8UC1
    with pytest.raises(Exception):
        ingest_device(capi.default_ingest(roi_x0=5, roi_y0=5, roi_x1=30, roi_y1=10), f)  # ROI outside the frame
    with pytest.raises(Exception):
        ingest_device(capi.default_ingest(resize_percent=1), f)  # nothing left


def test_ingest_full_size_1080p():
    import torch
    frames = synth.s_surv(2, 1080, 1920, seed=5, device="cuda")
    for kw in (dict(resize_percent=50, flip=1, gaussian_blur=1), dict(flip=1, roi_x0=100, roi_y0=60, roi_x1=1700, roi_y1=1000, gaussian_blur=1), dict(resize_percent=75)):
        cfg = capi.default_ingest(**kw)
        got = ingest_device(cfg, frames).cpu().numpy()
        host = frames.cpu().numpy()
        for k in range(2):
            assert np.array_equal(got[k], pyoracle.ingest(cfg, host[k])), kw


@pytest.mark.parametrize("kw", [dict(flip=1, roi_x0=6, roi_y0=4, roi_x1=86, roi_y1=68), dict(flip=1), dict(resize_percent=50, gaussian_blur=1), dict(gaussian_blur=1, flip=1)])
@pytest.mark.parametrize("algo", [capi.MOG2, capi.FRAME_DIFF])
def test_engine_with_ingest_matches_oracle_on_prepared_frames(kw, algo, golden_frames):
    """bgs_set_ingest: bgs_process takes the raw frame; masks / backgrounds equal the oracle run on the oracle-prepared frames.
    Flip + ROI alone ride on the staging copy (no device work), the others run bgs_ingest_device between upload and model kernel."""
    cfg = capi.default_ingest(**kw)
    eng, orc = Engine(algo), pyoracle.Oracle(algo)
    eng.set_ingest(cfg)
    for t, f in enumerate(golden_frames[:8]):
        fg, bg = eng.process(f)
        ofg, obg = orc.process(pyoracle.ingest(cfg, f))
        assert (fg is None) == (ofg is None) and (bg is None) == (obg is None), t
        if fg is not None:
            assert np.array_equal(fg, ofg), (t, kw)
        if bg is not None:
            assert np.array_equal(bg, obg), (t, kw)
    with pytest.raises(Exception):
        eng.set_ingest(capi.default_ingest())  # after the first frame
    eng.close()
