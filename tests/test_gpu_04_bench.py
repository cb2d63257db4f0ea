"""bench.py itself, on a small batch so that it takes seconds: the JSON contract of the line the driver parses, and the N > 1 code
path's RCCL calls (process group, per-step packed-mask gather, barrier, max-reduce) with the one rank a one-GPU box has."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--streams", "2", "--settle", "4", "--sustain", "4", *extra],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _bench("--no-pmc")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "Mpixels/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["kernel_launches"] == 4
    assert r["achieved"] > 0 and d["value"] > 0 and abs(d["value"] - 2 * 1920 * 1080 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.01
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mpixels/s" and "sample" in c
    assert d["clip"]["T4"]["frames_per_launch"] == 4 and d["clip"]["T8"]["kernel"] == "mog2_clip_kernel"
    # round 4: the line explains itself - this box's own copy rate through both allocation schemes, the headline against it, the whole
    # model streamed as the placement witness, the filter's worst-case scene, the quiet scene with the slot layout's byte model
    cal = d["calibration"]
    assert cal["copy_GBps_plain"] > 0 and cal["copy_GBps_chunked"] > 0 and cal["bytes"] >= 2 * 1920 * 1080 * 112
    assert "clocks" in cal and ("error" in cal["clocks"] or cal["clocks"]["samples"] >= 0)  # engine / memory clock during the sustained launches where sysfs shows them
    assert abs(r["frac_of_box_copy"] - r["achieved"] / cal["copy_GBps_chunked"]) < 1e-3
    assert d["model_placement"]["dense_launch"]["kernel_ms"] > 0 and d["model_placement"]["dense_launch"]["bytes_per_pixel"] == 228
    assert d["s_dense"]["mean_live_modes_stream0"] > 4.0 and d["s_dense"]["kernel_ms"] > 0
    sv = d["s_surv"]["default"]
    assert 1.0 <= sv["mean_live_modes_stream0"] < 3.0 and abs(sv["bytes_model_per_pixel"] - (22 + 24 * sv["mean_live_modes_stream0"])) < 0.1
    hp = d["host_path"]
    assert hp["pcie_calibration"]["registered_pageable"]["h2d_GBps"] > 0
    assert hp["registered_buffers"]["diag"]["pinned_input"] == 1 and hp["registered_buffers"]["diag"]["pinned_mask"] == 1 and hp["staged"]["diag"]["pinned_input"] == 0
    assert hp["submit_wait_8_cameras_registered_buffers"]["diag"]["pinned_input"] == 8 and hp["submit_wait_8_cameras_registered_buffers"]["diag"]["register_calls_inside_timed_rounds"] == 0
    assert hp["submit_wait_8_cameras_registered_arena"]["diag"]["arenas"] == 2 and hp["submit_wait_8_cameras_registered_arena"]["diag"]["register_calls"] == 2
    assert hp["submit_wait_8_cameras_registered_arena"]["diag"]["cpu_ms_staging_per_frame"] == 0


def test_bench_rccl_selftest_gathers_what_the_kernel_wrote():
    d = _bench("--rccl-selftest", "--main-only")
    assert d["config"]["mask_gather"].startswith("RCCL SELF-TEST")
    s = d["rccl_selftest_gather_matches_kernel_output"]
    assert s["gather_equals_kernel_output"] is True and s["nonzero_words"] > 0 and s["words"] == 2 * 1920 * 1080 // 64
    pr = d["per_rank"]  # N > 1 diagnostics, present whenever the collective path runs
    assert len(pr["kernel_avg_ms"]) == 1 and pr["kernel_avg_ms"][0] > 0 and len(pr["gather_wait_ms_per_step"]) == 1 and pr["gather_wait_ms_per_step"][0] >= 0


def test_bench_native_node_selftest_runs_the_gather_through_libbgs_node():
    """--native-node: the step goes through bgs_node_step_device (rank form, ncclCommInitRank with a broadcast id); with one rank the
    block is sent to itself and received from itself through RCCL every step."""
    d = _bench("--rccl-selftest", "--native-node", "--main-only")
    assert "libbgs_node" in d["config"]["mask_gather"]
    s = d["rccl_selftest_gather_matches_kernel_output"]
    assert s["words"] == 2 * 1920 * 1080 // 64 and s["nonzero_words"] > s["words"] // 10  # an inverted frame is foreground in most places (tests/test_gpu_07_node.py holds the gathered words against the oracle)
    assert d["roofline"]["kernel_launches"] == 4 and d["value"] > 0 and d["per_rank"]["gather_wait_ms_per_step"][0] >= 0


def test_bench_rehearsal_two_ranks_reports_per_rank_diagnostics():
    """The N > 1 control flow with two ranks on the one GPU of this box (gloo, masks through host copies - not a benchmark): the line
    must carry what a first scaling run needs to be diagnosed - every rank's kernel time, its wall time per step, and the time
    MaskGather.next_buffer() blocked on the gather."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--streams", "2", "--settle", "4", "--sustain", "4", "--rehearse", "--main-only"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["mask_gather"].startswith("REHEARSAL")
    pr = d["per_rank"]
    for k in ("kernel_avg_ms", "gather_wait_ms_per_step", "ms_per_step_local"):
        assert len(pr[k]) == 2 and all(v >= 0 for v in pr[k]), k
    assert all(v > 0 for v in pr["kernel_avg_ms"]) and max(pr["ms_per_step_local"]) <= d["ms_per_step"] * 1.001
    assert abs(d["value"] - 2 * 2 * 1920 * 1080 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.01  # whole-job aggregate over both ranks
