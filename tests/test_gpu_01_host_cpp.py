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
