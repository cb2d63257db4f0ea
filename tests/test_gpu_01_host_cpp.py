"""The C++ host layer above the C ABI (tracking_amd/host): IBGS / FrameProcessor mirror + ./config/<Class>.xml handling.
CPU part: it compiles with plain g++, writes the reference's XML files with the reference's defaults, and a box without a
GPU gets ONE std::exception (the reference's CV_Assert -> cv::Exception path), not a fallback.
GPU part: the demo harness's masks equal the oracle's for every class FrameProcessor fans out to."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tracking_amd", "host")
DEMO = os.path.join(ROOT, "tracking_amd", "lib", "bgs_demo")

CLASSES = ["FrameDifferenceBGS", "StaticFrameDifferenceBGS", "WeightedMovingMeanBGS", "WeightedMovingVarianceBGS",
           "MixtureOfGaussianV1BGS", "MixtureOfGaussianV2BGS", "AdaptiveBackgroundLearning", "AdaptiveSelectiveBackgroundLearning",
           "GMG", "DPAdaptiveMedianBGS", "DPGrimsonGMMBGS", "DPZivkovicAGMMBGS", "DPMeanBGS", "DPWrenGABGS", "SigmaDeltaBGS", "SuBSENSEBGS", "LOBSTERBGS"]


@pytest.fixture(scope="module")
def demo():
    subprocess.run(["make", "-s", "-C", HOST], check=True)
    assert os.path.exists(DEMO)
    return DEMO


def write_fp_config(cfg_dir, enabled, tictoc=""):
    os.makedirs(cfg_dir, exist_ok=True)
    with open(os.path.join(cfg_dir, "FrameProcessor.xml"), "w") as f:
        f.write('<?xml version="1.0"?>\n<opencv_storage>\n<tictoc>"%s"</tictoc>\n<enablePreProcessor>1</enablePreProcessor>\n' % tictoc)
        for c in CLASSES:
            f.write("<enable%s>%d</enable%s>\n" % (c, 1 if c in enabled else 0, c))
        f.write("</opencv_storage>\n")


def run_demo(demo, workdir, frames):
    raw = os.path.join(workdir, "frames.raw")
    frames.tofile(raw)
    n, rows, cols = frames.shape[:3]
    return subprocess.run([demo, raw, str(rows), str(cols), str(n), os.path.join(workdir, "out")], cwd=workdir, capture_output=True, text=True)


def test_host_layer_without_gpu_fails_with_one_exception(demo, tmp_path, golden_frames):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    write_fp_config(str(tmp_path / "config"), ["MixtureOfGaussianV2BGS"])
    r = run_demo(demo, str(tmp_path), golden_frames[:3])
    assert r.returncode == 1
    assert "std::exception:" in r.stdout and "no HIP device" in r.stdout
    assert "MixtureOfGaussianV2BGS()" in r.stdout  # the reference's ctor banner
    # saveConfig() ran on the first process() call and wrote the reference's keys with the reference's defaults
    xml = (tmp_path / "config" / "MixtureOfGaussianV2BGS.xml").read_text()
    assert xml.startswith('<?xml version="1.0"?>\n<opencv_storage>\n')
    for frag in ("<alpha>0.05</alpha>", "<enableThreshold>1</enableThreshold>", "<threshold>15</threshold>", "<showOutput>1</showOutput>"):
        assert frag in xml, xml


def test_default_frameprocessor_config_matches_reference(demo, tmp_path, golden_frames):
    """No ./config/FrameProcessor.xml: the constructor writes one with only FrameDifferenceBGS enabled (config/FrameProcessor.xml:6)."""
    os.makedirs(tmp_path / "config")
    run_demo(demo, str(tmp_path), golden_frames[:2])
    xml = (tmp_path / "config" / "FrameProcessor.xml").read_text()
    assert "<enableFrameDifferenceBGS>1</enableFrameDifferenceBGS>" in xml
    assert "<enableMixtureOfGaussianV2BGS>0</enableMixtureOfGaussianV2BGS>" in xml
    assert '<tictoc>""</tictoc>' in xml


@pytest.mark.gpu
def test_demo_masks_equal_oracle_for_every_class(demo, tmp_path, golden_frames):
    from oracle import pyoracle
    from tracking_amd import capi
    algo = dict(zip(CLASSES, [capi.FRAME_DIFF, capi.STATIC_FRAME_DIFF, capi.WMM, capi.WMV, capi.MOG1, capi.MOG2, capi.ABL, capi.ASBL,
                              capi.GMG, capi.DP_ADAPTIVE_MEDIAN, capi.DP_GRIMSON_GMM, capi.DP_ZIVKOVIC_AGMM, capi.DP_MEAN, capi.DP_WREN_GA, capi.SIGMA_DELTA, capi.SUBSENSE, capi.LOBSTER]))
    write_fp_config(str(tmp_path / "config"), CLASSES, tictoc="MixtureOfGaussianV2BGS")
    # a non-default per-class config must be honoured too (the reference re-reads it every frame)
    (tmp_path / "config" / "DPGrimsonGMMBGS.xml").write_text(
        '<?xml version="1.0"?>\n<opencv_storage>\n<threshold>16.</threshold>\n<alpha>0.05</alpha>\n<gaussians>4</gaussians>\n<showOutput>0</showOutput>\n</opencv_storage>\n')
    (tmp_path / "config" / "WeightedMovingVarianceBGS.xml").write_text(
        '<?xml version="1.0"?>\n<opencv_storage>\n<enableWeight>0</enableWeight>\n<enableThreshold>1</enableThreshold>\n<threshold>9</threshold>\n<showOutput>0</showOutput>\n</opencv_storage>\n')
    frames = golden_frames[:10]
    r = run_demo(demo, str(tmp_path), frames)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("MixtureOfGaussianV2BGS\ttime(sec):") == len(frames)  # tic/toc line format of FrameProcessor.cpp:493
    assert "FrameProcessor: 6 byte-stream classes share one launch per frame" in r.stdout  # FD, SFD, WMM, WMV, ABL, SigmaDelta as one bgs_group
    n, rows, cols = frames.shape[:3]
    for c in CLASSES:
        got = np.fromfile(str(tmp_path / ("out.%s.raw" % c)), np.uint8).reshape(n, rows, cols)
        p = capi.default_params(algo[c])
        if c == "WeightedMovingVarianceBGS":
            p.enable_weight, p.threshold = 0, 9
        if c == "DPGrimsonGMMBGS":
            p.dp_threshold, p.dp_alpha, p.dp_gaussians = 16.0, 0.05, 4
        o = pyoracle.Oracle(algo[c], params=p)
        for t in range(n):
            fg, _ = o.process(frames[t])
            if fg is None:
                assert (got[t] == 7).all(), (c, t)  # output left untouched (still empty) -> demo writes the 0x07 marker
            else:
                assert np.array_equal(got[t], fg), (c, t)
    # the same run with every class on its own engine (BGS_HOST_NO_GROUP=1): identical files
    raw = {c: (tmp_path / ("out.%s.raw" % c)).read_bytes() for c in CLASSES}
    r2 = subprocess.run([demo, str(tmp_path / "frames.raw"), str(rows), str(cols), str(n), str(tmp_path / "out")], cwd=str(tmp_path), capture_output=True, text=True,
                        env=dict(os.environ, BGS_HOST_NO_GROUP="1"))
    assert r2.returncode == 0 and "share one launch" not in r2.stdout
    for c in CLASSES:
        assert (tmp_path / ("out.%s.raw" % c)).read_bytes() == raw[c], c


@pytest.mark.gpu
@pytest.mark.parametrize("utype,algo_name", [(36, "SUBSENSE"), (5, "MOG2"), (11, "DP_ZIVKOVIC_AGMM"), (37, "LOBSTER"), (0, "FRAME_DIFF")])
def test_ustc_bgs_type_table(demo, tmp_path, golden_frames, utype, algo_name):
    """USTC_BGS(type).Process / GetMask (ustc_src/ustc_bgs.cpp): type 36 is what trackingMain.cpp builds.  GetMask() hands out the
    last mask; for FrameDifference the first frame leaves img_mask empty (written as the 0x07 marker)."""
    from oracle import pyoracle
    from tracking_amd import capi
    os.makedirs(tmp_path / "config")
    frames = golden_frames[:8]
    raw = os.path.join(str(tmp_path), "frames.raw")
    frames.tofile(raw)
    n, rows, cols = frames.shape[:3]
    r = subprocess.run([demo, raw, str(rows), str(cols), str(n), os.path.join(str(tmp_path), "out"), str(utype)], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(str(tmp_path / "out.ustc.raw"), np.uint8).reshape(n, rows, cols)
    o = pyoracle.Oracle(getattr(capi, algo_name))
    last = None
    for t in range(n):
        fg, _ = o.process(frames[t])
        last = fg if fg is not None else last
        if last is None:
            assert (got[t] == 7).all()
        else:
            assert np.array_equal(got[t], last), t


@pytest.mark.gpu
def test_ustc_bgs_rejects_types_outside_the_path(demo, tmp_path, golden_frames):
    os.makedirs(tmp_path / "config")
    raw = os.path.join(str(tmp_path), "frames.raw")
    golden_frames[:2].tofile(raw)
    r = subprocess.run([demo, raw, "80", "96", "2", os.path.join(str(tmp_path), "out"), "23"], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 1 and "outside the package_bgs hot path" in r.stdout


def _oracle_blobs(mask, min_w=5, min_h=5, connectivity=8):
    """(boxes[k] as pyoracle.BOX_DTYPE, moments int64 [k][4]) of the oracle's components of `mask`, smaller ones dropped."""
    from oracle import pyoracle
    labels, boxes, n = pyoracle.components(mask, connectivity)
    keep = [b for b in boxes if b["w"] >= min_w and b["h"] >= min_h]
    ys, xs = np.nonzero(labels >= 0)
    roots = labels[ys, xs]
    mom = []
    for b in keep:
        sel = roots == b["root"]
        x, y = xs[sel].astype(np.int64), ys[sel].astype(np.int64)
        mom.append([x.sum(), y.sum(), (x * x).sum(), (y * y).sum()])
    return keep, np.array(mom, np.int64).reshape(-1, 4)


@pytest.mark.gpu
def test_last_mask_blobs_match_oracle_components(golden_frames):
    """N2, C ABI: bgs_last_mask_blobs on the device copy of the mask == the oracle's connected components of the oracle's mask
    (rectangles, areas, first pixels, coordinate moments), with and without the mask also being returned to the host."""
    from oracle import pyoracle
    from tracking_amd import Engine, capi
    eng, orc = Engine(capi.MOG2), pyoracle.Oracle(capi.MOG2)
    seen_blobs = 0
    for t, f in enumerate(golden_frames[:12]):
        ofg, _ = orc.process(f)
        if t % 2:
            flags = eng.process_mask_only_on_device(f)  # fg = NULL: nothing but rectangles crosses PCIe
            assert flags & capi.FG_VALID
        else:
            fg, _ = eng.process(f)
            assert np.array_equal(fg, ofg)
        for conn, mw, mh in ((8, 5, 5), (4, 0, 0), (8, 2, 3)):
            boxes, mom, n = eng.last_mask_blobs(connectivity=conn, min_w=mw, min_h=mh, max_boxes=4096)
            want, wmom = _oracle_blobs(ofg, mw, mh, conn)
            assert n == len(want), (t, conn, n, len(want))
            assert np.array_equal(boxes, np.array([[b[k] for k in ("x", "y", "w", "h", "area", "root")] for b in want], np.int32).reshape(-1, 6)), (t, conn)
            assert np.array_equal(mom, wmom), (t, conn)
            seen_blobs += n
    assert seen_blobs > 20
    # truncation: count still says how many there are
    boxes, _, n = eng.last_mask_blobs(connectivity=4, max_boxes=3)
    assert n > 3 and len(boxes) == 3
    eng.close()


@pytest.mark.gpu
def test_last_mask_blobs_keeps_a_large_blob_behind_more_speckles_than_the_scratch_holds():
    """1 600 one-pixel components in the top rows (more than the 1 024 boxes the device scratch starts with), then one 30x20 block:
    with min_w = min_h = 5 and max_boxes = 256 (what HipFGDetector::GetBlobs asks for) the block must come back - the size filter
    runs after ALL components were kept, not on the first 1 024 in raster order."""
    from tracking_amd import Engine, capi
    H, W = 96, 160
    f0 = np.zeros((H, W, 3), np.uint8)
    f1 = f0.copy()
    f1[0:40:2, 0:W:2] = 255
    f1[60:80, 100:130] = 200
    eng = Engine(capi.FRAME_DIFF)
    eng.process(f0)
    fg, _ = eng.process(f1)
    assert int((fg != 0).sum()) == 20 * 80 + 20 * 30
    boxes, _, n = eng.last_mask_blobs(connectivity=8, min_w=5, min_h=5, max_boxes=256)
    assert n == 1 and boxes[0, :5].tolist() == [100, 60, 30, 20, 600], (n, boxes[:2])
    boxes, _, n = eng.last_mask_blobs(connectivity=8, min_w=0, min_h=0, max_boxes=4096)  # unfiltered: every speckle and the block
    assert n == 1601 and boxes[-1, :5].tolist() == [100, 60, 30, 20, 600]
    boxes, _, n = eng.last_mask_blobs(connectivity=8, min_w=0, min_h=0, max_boxes=8)  # truncated output, true count
    assert n == 1601 and len(boxes) == 8
    eng.close()


@pytest.mark.gpu
def test_last_mask_blobs_needs_a_valid_mask(golden_frames):
    from tracking_amd import Engine, capi
    eng = Engine(capi.FRAME_DIFF, n_streams=2)
    eng.process(golden_frames[0])  # warm-up frame: outputs untouched, no mask
    with pytest.raises(Exception):
        eng.last_mask_blobs()
    eng.process(golden_frames[1])
    eng.last_mask_blobs()
    eng.process(golden_frames[0], stream=1)
    with pytest.raises(Exception):
        eng.last_mask_blobs(stream=0)  # the device mask now belongs to stream 1 (and that was a warm-up frame)
    eng.close()


@pytest.mark.gpu
def test_mask_blobs_batch_device_vs_oracle():
    import torch
    from tracking_amd.engine import mask_blobs_batch_device
    rng = np.random.default_rng(5)
    masks = np.zeros((3, 120, 200), np.uint8)
    for k in range(3):
        for _ in range(25):
            y, x = rng.integers(0, 110), rng.integers(0, 180)
            masks[k, y:y + rng.integers(1, 30), x:x + rng.integers(1, 40)] = 255
        masks[k][rng.random(masks[k].shape) < 0.01] = 255
    boxes, mom, off = mask_blobs_batch_device(torch.from_numpy(masks).cuda(), 8)
    boxes, mom = boxes.cpu().numpy(), mom.cpu().numpy()
    for k in range(3):
        want, wmom = _oracle_blobs(masks[k], 0, 0, 8)
        got = boxes[off[k]:off[k + 1]]
        assert len(got) == len(want)
        assert np.array_equal(got, np.array([[b[q] for q in ("x", "y", "w", "h", "area", "root")] for b in want], np.int32).reshape(-1, 6))
        assert np.array_equal(mom[off[k]:off[k + 1]], wmom)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["box", "moments"])
def test_hipfgdetector_blob_list(demo, tmp_path, golden_frames, mode):
    """N2, C++ mirror: HipFGDetector(5).Process + GetBlobs through the demo harness; the printed CvBlob {x, y, w, h, ID} list of every
    frame equals what blob_convert.h's formulas give on the oracle's components of the oracle's mask (regions >= 5 x 5, raster order,
    IDs counting up; the harness prints the list back to front like ustc_src/trackingMain.cpp:184-190)."""
    from oracle import pyoracle
    from tracking_amd import capi
    os.makedirs(tmp_path / "config")
    frames = golden_frames[:10]
    raw = os.path.join(str(tmp_path), "frames.raw")
    frames.tofile(raw)
    n, rows, cols = frames.shape[:3]
    r = subprocess.run([demo, raw, str(rows), str(cols), str(n), os.path.join(str(tmp_path), "out"), "5", mode], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.split("\n") if l.startswith("frame ") or l.startswith("pBlob")]
    o = pyoracle.Oracle(capi.MOG2)
    next_id, pos, total = 0, 0, 0
    for t in range(n):
        fg, _ = o.process(frames[t])
        want, wmom = _oracle_blobs(fg, 5, 5, 8)
        assert lines[pos] == "frame %d blobs %d" % (t, len(want)), (lines[pos], len(want))
        pos += 1
        exp = []
        for b, m in zip(want, wmom):
            if mode == "box":
                x, y, w, h = b["x"] + 0.5 * b["w"], b["y"] + 0.5 * b["h"], float(b["w"]), float(b["h"])
            else:
                a = float(b["area"])
                mx, my = m[0] / a, m[1] / a
                x, y = mx, my
                w, h = 4 * np.sqrt(max(m[2] / a - mx * mx, 0)), 4 * np.sqrt(max(m[3] / a - my * my, 0))
            exp.append((x, y, w, h, next_id))
            next_id += 1
        for e in reversed(exp):
            vals = lines[pos].replace("pBlob x,y,w,h,id is ", "").split(" , ")
            pos += 1
            got = [float(v) for v in vals[:4]]
            assert np.allclose(got, e[:4], rtol=1e-5, atol=1e-4), (t, got, e)
            assert int(vals[4]) == e[4]
        total += len(exp)
    assert total > 5


@pytest.mark.gpu
def test_frameprocessor_with_videocapture_prep_and_preprocessor(demo, tmp_path, golden_frames):
    """N3 through the C++ mirror: ./config/VideoCapture.xml (resize 50 %, flip, ROI) and ./config/PreProcessor.xml (gaussianBlur) are
    honoured in front of every class: the demo's masks equal the oracle's masks of the oracle-prepared frames."""
    from oracle import pyoracle
    from tracking_amd import capi
    cfg_dir = tmp_path / "config"
    write_fp_config(str(cfg_dir), ["MixtureOfGaussianV2BGS", "FrameDifferenceBGS"])
    (cfg_dir / "VideoCapture.xml").write_text('<?xml version="1.0"?>\n<opencv_storage>\n<input_resize_percent>50</input_resize_percent>\n<enableFlip>1</enableFlip>\n'
                                               '<use_roi>1</use_roi>\n<roi_defined>1</roi_defined>\n<roi_x0>2</roi_x0>\n<roi_y0>3</roi_y0>\n<roi_x1>44</roi_x1>\n<roi_y1>35</roi_y1>\n</opencv_storage>\n')
    (cfg_dir / "PreProcessor.xml").write_text('<?xml version="1.0"?>\n<opencv_storage>\n<equalizeHist>0</equalizeHist>\n<gaussianBlur>1</gaussianBlur>\n<enableShow>0</enableShow>\n</opencv_storage>\n')
    frames = golden_frames[:8]
    r = run_demo(demo, str(tmp_path), frames)
    assert r.returncode == 0, r.stdout + r.stderr
    cap = capi.default_ingest(resize_percent=50, flip=1, roi_x0=2, roi_y0=3, roi_x1=44, roi_y1=35)
    pre = capi.default_ingest(gaussian_blur=1)
    prepared = [pyoracle.ingest(pre, pyoracle.ingest(cap, f)) for f in frames]
    rows, cols = prepared[0].shape[:2]
    assert (rows, cols) == (32, 42)
    for c, algo in (("MixtureOfGaussianV2BGS", capi.MOG2), ("FrameDifferenceBGS", capi.FRAME_DIFF)):
        got = np.fromfile(str(tmp_path / ("out.%s.raw" % c)), np.uint8).reshape(len(frames), rows, cols)
        o = pyoracle.Oracle(algo)
        for t, f in enumerate(prepared):
            fg, _ = o.process(f)
            if fg is None:
                assert (got[t] == 7).all(), (c, t)
            else:
                assert np.array_equal(got[t], fg), (c, t)
