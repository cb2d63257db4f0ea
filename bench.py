#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: Mpixels/s (and concurrent 1080p30 streams) of MixtureOfGaussianV2BGS.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S] [--input sat|surv]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...     (one rank per GPU, RCCL)

A step = one frame of every stream on this rank's GPU through bgs_process_batch_device (ONE kernel launch over
streams x pixels, frames already resident in HBM).  Streams shard across ranks with no data-path collective;
with N > 1 the bit-packed masks are gathered to rank 0 every step (the one real exchange step),
overlapped with the next step's kernel.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROWS, COLS, CH = 1080, 1920, 3
SURVEY_BYTES_PER_PIXEL = 206  # SURVEY.md §8(d), the reference's formulation (modes physically sorted): r 3 frame + 1 nmodes + 5*8 {w,var} + 5*12 mu ; w 5*8 + 5*12 + 1 nmodes + 1 mask
# What THIS formulation must move per pixel and frame on the all-modes-live, well-separated input S_sat (kernel_mog2.h:
# weights by rank, {var, mean} records in fixed slots, a 16-bit rank->slot word, 2-byte summaries that rule modes out without their record):
#   read  3 frame + 2 meta + 5*4 weights + 5*2 summaries + 16 (the ONE record the summaries cannot rule out) = 51
#   write 5*4 weights + 16 (that record, updated) + 2 meta (the order changes every frame) + 1 mask = 39
#   (+ 2 when the record's summary no longer covers it and is rewritten)
# (round 3: 4-byte summaries, 100 B/pixel)
BYTES_PER_PIXEL = 90
HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured achievable)


def make_source(kind, streams, device, seed0):
    """A stateful frame source for `streams` cameras of this GPU (tools/synth.py): FRESH noise on every frame it produces.
    Stream block seeds follow SURVEY.md 8d (config 5: seeds 1234+i / 4321+i): one generator per rank, seeded with the base + its first global stream."""
    from tools import synth
    cls = {"sat": synth.SatStreams, "surv": synth.SurvStreams, "dense": synth.DenseStreams}[kind]
    return cls(streams, ROWS, COLS, seed0=seed0, device=device)


MODEL_BYTES_PER_PIXEL = 112  # kernel_mog2.h: 20 weights + 10 summaries + 80 records + 2 meta


def calibrate(local, streams):
    """`calibration`: this box's own yardstick, taken in this process BEFORE the model exists - a float4 copy over the model's size
    through (a) one plain hipMalloc and (b) one virtual range backed by 256 MiB physical chunks, the construction the model itself uses
    (DESIGN.md 6.2).  Boxes of the pool differ by 5-9 % on every HBM-bound kernel; `roofline.frac_of_box_copy` prices the headline
    kernel against (b), so the line says whether a slow number is a slow box."""
    from tracking_amd import capi
    nbytes = (streams * ROWS * COLS + 255) // 256 * 256 * MODEL_BYTES_PER_PIXEL
    out = {"bytes": nbytes, "kernel": "float4 copy, bytes/2 read + bytes/2 written per launch, mean of 5 launches after 2 warm-ups (bgs_calibrate_copy)"}
    try:  # what the box is: boxes of the pool differ by 5-12 % on the same kernel with the same clock readings
        import torch
        pr = torch.cuda.get_device_properties(local)
        out["device"] = {"name": pr.name, "arch": getattr(pr, "gcnArchName", None), "compute_units": pr.multi_processor_count, "memory_GiB": round(pr.total_memory / 2**30, 1),
                         "l2_MiB": round(getattr(pr, "L2_cache_size", 0) / 2**20, 1), "max_engine_clock_MHz": round(getattr(pr, "clock_rate", 0) / 1e3, 0) or None}
    except Exception as ex:  # noqa: BLE001 - diagnostics only
        out["device"] = {"error": repr(ex)}
    for label, chunk in (("copy_GBps_plain", 0), ("copy_GBps_chunked", 256), ("copy_GBps_plain_again", 0)):
        try:
            out[label] = round(capi.calibrate_copy(local, nbytes, chunk), 1)
        except Exception as ex:  # noqa: BLE001 - diagnostics only
            out[label] = None
            out[label + "_error"] = repr(ex)
    try:  # what WRITES alone reach (the yardstick of kernels that only store: SuBSENSE's model initialisation): hipMemsetAsync over the same size
        import torch
        buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda:%d" % local)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            buf.zero_()
        a.record()
        for _ in range(5):
            buf.zero_()
        b.record()
        torch.cuda.synchronize()
        out["fill_GBps_plain"] = round(nbytes * 5 / (a.elapsed_time(b) * 1e-3) / 1e9, 1)
        del buf
        torch.cuda.empty_cache()
    except Exception as ex:  # noqa: BLE001 - diagnostics only
        out["fill_GBps_plain"] = None
        out["fill_GBps_plain_error"] = repr(ex)
    return out


from tools.clocks import ClockSampler  # noqa: E402 - engine / memory clock and board power from sysfs while a leg runs


def host_cores():
    """Threads this job may really use: the cgroup CPU quota when there is one, else the affinity mask capped at the
    GPU box's per-GPU share of 16 (an OpenMP team larger than the quota only thrashes)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    env = os.environ.get("BGS_BENCH_THREADS")
    return int(env) if env else min(n, 16)


def cpu_baseline(frames, kind, budget_s=16.0):
    """The oracle (CPU restatement, kind='port') timed on this box's host cores on a bounded sample of the same
    workload: stream 0's frames ([period][H][W][3] numpy), model warmed for 2 periods, then as many 1080p frames as fit in ~budget_s."""
    from oracle import pyoracle
    from tracking_amd import capi
    cores = host_cores()
    period = frames.shape[0]
    res = {}
    for label, threads, share in (("all", cores, 0.7), ("one", 1, 0.3)):
        orc = pyoracle.Oracle(capi.MOG2, threads=threads)
        t = 0
        for _ in range(2 * period if kind == "sat" else period):
            orc.process(frames[t % period], want_bg=False)
            t += 1
        t0 = time.perf_counter()
        orc.process(frames[t % period], want_bg=False)
        t += 1
        one = time.perf_counter() - t0
        n = int(max(3, min(400, budget_s * share / max(one, 1e-4))))
        t0 = time.perf_counter()
        for _ in range(n):
            orc.process(frames[t % period], want_bg=False)
            t += 1
        dt = time.perf_counter() - t0
        res[label] = (n * ROWS * COLS / dt / 1e6, n, threads)
        orc.close()
    return {
        "value": round(res["all"][0], 2), "unit": "Mpixels/s", "cores": res["all"][2], "kind": "port",
        "sample": "%d frames 1920x1080x3 of stream 0 (same synthetic clip, model warmed), oracle MOG2 update+classify+threshold, rows split over %d OpenMP threads" % (res["all"][1], res["all"][2]),
        "single_thread_value": round(res["one"][0], 2),
        "note": "OpenCV 2.4's own MOG2 is absent from the reference tree and the image; this is the build's CPU restatement (oracle/bgs_oracle.c)",
    }


HOSTPATH_FIELDS = ("pinned_input", "pinned_mask", "pinned_background", "refused_input", "refused_mask", "refused_background", "register_calls", "register_ms_total",
                   "unregister_calls", "frames", "h2d_bytes", "d2h_bytes", "cpu_ms_staging_in", "cpu_ms_staging_out", "arenas")


def _hostpath_diag(e):
    """an engine's host-path counters (bgs_get_state "hostpath"): which buffers are page-locked in place, what registering them cost,
    how long the CPU copied - so that a slow leg names its own cause"""
    v = e.get_state("hostpath", (len(HOSTPATH_FIELDS),), np.float64)
    d = {k: (round(float(x), 3) if "ms" in k else int(x)) for k, x in zip(HOSTPATH_FIELDS, v)}
    fr = max(1, d["frames"])
    d["cpu_ms_staging_per_frame"] = round((d["cpu_ms_staging_in"] + d["cpu_ms_staging_out"]) / fr, 4)
    return d


def host_leg(local, clip, frames_n=120, cameras_only=False):
    """The drop-in call today's IBGS::process callers make - bgs_process, host buffers, one 1080p stream, synchronous - through the
    ctypes binding: staged (default: every image goes through the engine's pinned buffers) and with BGS_OPT_HOST_REGISTER (the caller's
    frame and mask buffers stay allocated, as OpenCV's capture loop keeps them: page-locked once, DMA'd in place).  PCIe-inclusive,
    never `value`.  clip: [period][H][W][3] numpy, stream 0's frames."""
    from tracking_amd import Engine, capi
    period = clip.shape[0]
    out = {"note": "bgs_process (host buffers, 1 x 1920x1080x3 stream, mask only, synchronous) through the ctypes binding; PCIe-inclusive; never `value`; "
                   "`diag` = the engine's own counters after the leg (buffers page-locked in place / refused per role, hipHostRegister calls and ms, CPU ms in staging copies)"}
    fb = ROWS * COLS * CH
    if cameras_only:
        out = {}
    pcie = {}
    for label, reg in (("pinned_hipHostMalloc", 0), ("registered_pageable", 1)):
        try:
            u, d, r = capi.calibrate_pcie(local, fb, reg)
            pcie[label] = {"h2d_GBps": round(u, 2), "d2h_GBps": round(d, 2), "hipHostRegister_ms": round(r, 3)}
        except Exception as ex:  # noqa: BLE001
            pcie[label] = {"error": repr(ex)}
    out["pcie_calibration"] = dict(pcie, note="bgs_calibrate_pcie: 20 copies of one 1080p BGR frame (6.2 MB) each way; what the staged and the registered legs can at best reach on this box")
    for label, reg in (() if cameras_only else (("staged", 0), ("registered_buffers", 3))):
        e = Engine(capi.MOG2, device=local)
        e.set_option(capi.OPT_HOST_REGISTER, reg)
        frame = np.empty_like(clip[0])  # ONE frame buffer, refilled per frame: what cvQueryFrame's image is
        fg = np.empty((ROWS, COLS), np.uint8)
        for t in range(20):
            frame[...] = clip[t % period]
            e.process_into(frame, fg)
        fill = 0.0
        t0 = time.perf_counter()
        for t in range(frames_n):
            f0 = time.perf_counter()
            frame[...] = clip[t % period]  # stands for the decoder writing the next frame; not part of the call
            fill += time.perf_counter() - f0
            e.process_into(frame, fg)
        dt = time.perf_counter() - t0 - fill
        out[label] = {"ms_per_frame": round(dt / frames_n * 1e3, 4), "frames_per_s": round(frames_n / dt, 1), "mpixels_per_s": round(frames_n * ROWS * COLS / dt / 1e6, 1), "diag": _hostpath_diag(e)}
        e.close()
    # several cameras: bgs_submit / bgs_wait - every camera's frame queued on its own lane before any is collected.  Three ways of
    # holding the cameras' images: staged; each buffer page-locked on its own (16 registrations for 8 cameras); ONE arena for all input
    # frames and one for all masks, registered once each (bgs_host_arena)
    cams = 8
    for label, reg, arena in (("submit_wait_8_cameras_staged", 0, False), ("submit_wait_8_cameras_registered_buffers", 3, False), ("submit_wait_8_cameras_registered_arena", 0, True)):
        e = Engine(capi.MOG2, device=local, n_streams=cams)
        e.set_option(capi.OPT_HOST_REGISTER, reg)
        if arena:
            fr_all, fg_all = np.empty((cams, ROWS, COLS, CH), np.uint8), np.empty((cams, ROWS, COLS), np.uint8)
            for c in range(cams):
                fr_all[c] = clip[(3 * c) % period]
            frames, fgs = [fr_all[c] for c in range(cams)], [fg_all[c] for c in range(cams)]
            try:
                e.host_arena(fr_all), e.host_arena(fg_all)
            except Exception as ex:  # noqa: BLE001
                out[label] = {"error": repr(ex)}
                e.close()
                continue
        else:
            frames = [np.ascontiguousarray(clip[(3 * c) % period]) for c in range(cams)]
            fgs = [np.empty((ROWS, COLS), np.uint8) for _ in range(cams)]
        for t in range(12):
            for c in range(cams):
                e.submit(frames[c], fgs[c], None, stream=c)
            for c in range(cams):
                e.wait(stream=c)
        reg_before = _hostpath_diag(e)["register_calls"]
        rounds = max(8, frames_n // cams)
        t0 = time.perf_counter()
        for t in range(rounds):
            for c in range(cams):
                e.submit(frames[c], fgs[c], None, stream=c)
            for c in range(cams):
                e.wait(stream=c)
        dt = time.perf_counter() - t0
        n = rounds * cams
        diag = _hostpath_diag(e)
        diag["register_calls_inside_timed_rounds"] = diag["register_calls"] - reg_before
        out[label] = {"cameras": cams, "ms_per_frame": round(dt / n * 1e3, 4), "frames_per_s_aggregate": round(n / dt, 1), "mpixels_per_s": round(n * ROWS * COLS / dt / 1e6, 1),
                      "bus_GBps": round(n * (fb + ROWS * COLS) / dt / 1e9, 2), "diag": diag}
        e.close()
    return out


UPDATE_LAUNCHES = [0]  # bgs_process_batch_device calls on MOG2 engines so far = mog2_update_kernel dispatches, in order (the PMC child marks its legs by it)


def scene_leg(local, S, fg, kind, steps=100, sparse=3, warm=150, pool=None, marks=None):
    """The same engine on another scene, reported beside `value` (which stays S_sat):
      surv   SURVEY.md 8d's headline-streams input - static background + sensor noise + moving boxes, ~1.2 live modes per pixel;
      dense  the WORST case of round 3's summary filter - five modes 8 grey levels apart, all live, none of which a 4-byte summary can
             rule out (kernel_mog2.h mog2_reject needs ~13 levels in every channel): every record of every pixel is read.
    `warm` untimed launches on frames generated one by one, then `steps` timed launches over a pool of DISTINCT frames resident in HBM
    (fresh noise in every frame).  Returns (dict, pool) so that a second leg on the same scene reuses the frames."""
    from tracking_amd import Engine, capi
    dev = torch.device("cuda", local)
    eng = Engine(capi.MOG2, device=local, n_streams=S)
    eng.set_option(capi.OPT_MOG2_SPARSE, sparse)
    eng.set_geometry(ROWS, COLS, CH)
    src = make_source(kind, S, dev, 4321 if kind == "surv" else 777)
    cur = torch.empty((S, ROWS, COLS, CH), dtype=torch.uint8, device=dev)
    for _ in range(warm):
        eng.process_batch_device(src.into(cur), fg, None, None)
        UPDATE_LAUNCHES[0] += 1
    if pool is None:
        pool = src.pool(steps)
    torch.cuda.synchronize()
    eng.enable_kernel_timing(True)
    a = UPDATE_LAUNCHES[0]
    t0 = time.perf_counter()
    for t in range(steps):
        eng.process_batch_device(pool[t % pool.shape[0]], fg, None, None)
        UPDATE_LAUNCHES[0] += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if marks is not None:
        marks[kind if sparse == 3 else "%s_sparse%d" % (kind, sparse)] = [a, UPDATE_LAUNCHES[0]]
    ms, _, _ = eng.kernel_timing()
    nm = eng.get_state("nmodes", (ROWS * COLS,), np.uint8, stream=0)
    n_live = float(nm.mean())
    mpix = steps * S * ROWS * COLS / dt / 1e6
    # what the slot layout must move on such a scene (kernel_mog2.h, count / eager kernel): read 3 frame + 2 meta + n (4 weight + 16 record);
    # write 4 n weights (every live weight decays) + 16 (the ONE record a frame recomputes) + 1 mask (+ 2 meta when the order changes)
    model = 22.0 + 24.0 * n_live
    out = {"mpixels_per_s": round(mpix, 1), "streams_1080p30": round(mpix / (ROWS * COLS / 1e6) / 30.0, 1), "kernel_ms": round(ms, 4),
           "mean_live_modes_stream0": round(n_live, 3), "foreground_ratio": round(float((fg != 0).float().mean()), 4),
           "bytes_model_per_pixel": round(model, 1),
           "bytes_model_derivation": "slot layout, n = mean live modes: r 3 frame + 2 meta + n (4 weight + 16 record) ; w 4 n weights + 16 the one record a frame recomputes + 1 mask (+ 2 meta when the rank order changes) = 22 + 24 n (SURVEY.md 8d's `8 + 28 n` priced the reference's sorted array)",
           "bytes_model_GBps": round(model * S * ROWS * COLS / (ms * 1e-3) / 1e9, 1), "sparse_mode": sparse, "timed_launches": steps, "distinct_frames": int(pool.shape[0]),
           "note": "%s input, %d streams, fresh noise in every frame; BGS_OPT_MOG2_SPARSE=%d (1 eager: every record read, 2 count: only the modes a pixel has, 4 filter: summaries first, 3 = automatic [default])" % ("S_" + kind, S, sparse)}
    eng.close()
    return out, pool


CLIP_PMC_LAUNCHES = 10  # clip launches per T in the counter passes


PMC_LEG_LAUNCHES = 20   # timed launches of the S_surv / S_dense legs inside a counter pass


def pmc_traffic_live(streams, steps, warmup, inp, settle, timeout_s=300):
    """roofline.traffic (and the traffic of the clip, S_surv and S_dense legs) measured IN this run: two child passes of this script under
    `rocprofv3 --kernel-trace --pmc` (FETCH_SIZE, then WRITE_SIZE: they do not fit one pass).  A child runs the timed leg WITH THE SAME
    settle / warm-up as the parent (the counters belong to the model state that is timed), then CLIP_PMC_LAUNCHES clip launches of 4 and
    of 8 frames, then the S_surv and S_dense legs (PMC_LEG_LAUNCHES timed launches each), and writes which update-kernel dispatches
    belong to which leg into a side file.  Counters are corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts half of
    a wide streaming read; both are in KiB).  Children are ordinary subprocesses (never an exec from this GPU-holding process), each under
    a timeout; any failure returns None and the caller falls back to the committed constant.
    Returns ({leg: {hbm_bytes_per_launch, read_bytes, write_bytes}}, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process is itself being profiled"
    clip_want = {"clip4": "mog2_clip_kernel<4>", "clip8": "mog2_clip_kernel<8>"}
    acc = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="bgs_pmc_", dir="/tmp")
        marks_path = os.path.join(d, "marks.json")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
               sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup), "--streams", str(streams),
               "--input", inp, "--pmc-child", marks_path, "--sustain", "0", "--settle", str(settle), "--no-cpu-baseline"]
        try:
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = p.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, 9)  # exactly the process group this call started
                p.wait()
                shutil.rmtree(d, ignore_errors=True)
                return None, "%s pass timed out after %d s" % (ctr, timeout_s)
            rows = []
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == ctr]
            marks = json.load(open(marks_path)) if os.path.exists(marks_path) else {}
            shutil.rmtree(d, ignore_errors=True)
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            upd = [float(r["Counter_Value"]) for r in rows if "mog2_update" in r["Kernel_Name"]]
            if rc != 0 or "update" not in marks or len(upd) < marks["update"][1]:
                return None, "%s pass: rc=%s, %d update dispatches, marks %s" % (ctr, rc, len(upd), sorted(marks))
            for leg, (a, b) in marks.items():
                if b <= len(upd) and b > a:
                    acc.setdefault(leg, {})[ctr] = sum(upd[a:b]) / (b - a)
            for leg, pat in clip_want.items():
                vals = [float(r["Counter_Value"]) for r in rows if pat in r["Kernel_Name"]]
                if len(vals) >= CLIP_PMC_LAUNCHES:
                    acc.setdefault(leg, {})[ctr] = sum(vals[-CLIP_PMC_LAUNCHES:]) / CLIP_PMC_LAUNCHES
        except Exception as ex:  # noqa: BLE001 - diagnostics only, the bench line must still be printed
            shutil.rmtree(d, ignore_errors=True)
            return None, "%s pass failed: %r" % (ctr, ex)
    res = {}
    for leg, v in acc.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            rd, wr = v["FETCH_SIZE"] * 1024 * 2, v["WRITE_SIZE"] * 1024
            res[leg] = {"hbm_bytes_per_launch": int(rd + wr), "read_bytes": int(rd), "write_bytes": int(wr)}
    return res, ("measured in this run: two child passes under rocprofv3 --kernel-trace --pmc (FETCH_SIZE*1024*2 + WRITE_SIZE*1024; the child ages its model exactly like this "
                 "process: settle %d; mean over the %d timed dispatches of the update kernel, %d of each clip kernel, %d of the S_surv / S_dense legs)" % (settle, steps, CLIP_PMC_LAUNCHES, PMC_LEG_LAUNCHES))


def clip_leg(eng, pool, period, S, T, launches=40, warm=10):
    """Supplementary, never `value`: the same engine and saturated model through bgs_process_clip_device, T consecutive frames of
    every stream per launch with the model held in registers (file-fed video, or a live deployment that accepts T-1 frame
    times of latency).  Results are bit-identical to T single-frame steps (tests/test_gpu_03_clip.py)."""
    fgT = torch.empty((T, S, ROWS, COLS), dtype=torch.uint8, device=pool.device)
    starts = [t0 for t0 in range(0, period - T + 1, 5)]  # S_sat repeats every 5 frames: windows that start at multiples of 5 keep every level in play
    for i in range(warm):
        t0 = starts[i % len(starts)]
        eng.process_clip_device(pool[t0:t0 + T], T, fgT)
    torch.cuda.synchronize()
    eng.enable_kernel_timing(True)
    w0 = time.perf_counter()
    for i in range(launches):
        t0 = starts[i % len(starts)]
        eng.process_clip_device(pool[t0:t0 + T], T, fgT)
    torch.cuda.synchronize()
    wall = time.perf_counter() - w0
    ms, n, name = eng.kernel_timing()
    eng.enable_kernel_timing(False)
    px = S * ROWS * COLS * T
    # per launch and pixel: model read once (2 meta + 20 weights + 80 records) and written back once (20 weights + 16 per record a frame of the
    # clip updated: min(T, 5) on S_sat + 2 meta); per frame 3 B of input and 1 B of mask
    model = (102.0 + 20.0 + 16.0 * min(T, 5) + 2.0) / T + 4.0
    return {"frames_per_launch": T, "kernel": name, "launches": int(n), "kernel_avg_ms": round(ms, 4), "ms_per_frame_step": round(ms / T, 4),
            "mpixels_per_s": round(px * launches / wall / 1e6, 1), "frames_per_s": round(px * launches / wall / (ROWS * COLS), 1),
            "streams_1080p30": round(px * launches / wall / (ROWS * COLS) / 30.0, 1),
            "bytes_model_per_pixel_frame": round(model, 1), "added_latency_frames": T - 1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--streams", type=int, default=32, help="streams per GPU (config 5: 32)")
    ap.add_argument("--input", choices=["sat", "surv", "dense"], default="sat")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-bg", action="store_true", help="also produce the background image every frame (+3 B/px)")
    ap.add_argument("--main-only", action="store_true", help="only the timed leg (used under rocprofv3 so the last K launches are the timed ones)")
    ap.add_argument("--pmc-child", default="", metavar="MARKS.json", help="what the counter passes run: the timed leg, a few clip launches of 4 and 8 frames, short S_surv and S_dense legs; "
                    "writes which update-kernel dispatches belong to which leg into MARKS.json")
    ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE configs[2] / configs[3] block")
    ap.add_argument("--settle", type=int, default=340, help="untimed launches after model saturation and before the W warm-up steps: they age the model until the weights have equalised and every frame re-orders the modes (steady-state traffic)")
    ap.add_argument("--sustain", type=int, default=200, help="further launches after the timed K, reported as roofline.sustained")
    ap.add_argument("--series", default="", help="write the per-launch kernel durations (settle, warmup, timed, sustain) to this CSV")
    ap.add_argument("--rehearse", action="store_true", help="N > 1 control flow on one GPU: all ranks on cuda:0, gloo, masks via host (not a benchmark)")
    ap.add_argument("--rccl-selftest", action="store_true", help="one rank, but through the N > 1 code path: RCCL process group, packed-mask gather every step (to itself), barrier and max-reduce of the time - the collective calls on real hardware where only one GPU is available (not a scaling result)")
    ap.add_argument("--native-node", action="store_true", help="N > 1 (or --rccl-selftest): the gather through libbgs_node (include/bgs_node.h: ncclSend / ncclRecv issued by the C++ node driver, "
                    "double-buffered on a second HIP stream) instead of torch.distributed.gather; torch.distributed then only carries the 128-byte communicator id, the barriers and the max of the time")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic with rocprofv3 child passes (then the constant of profiles/pmc_traffic.json is reported)")
    ap.add_argument("--pool-frames", type=int, default=240, help="cap on the DISTINCT frames kept resident for the warm-up + timed steps (32 streams: 199 MB per frame); beyond it the pool is cycled")
    ap.add_argument("--px", type=int, default=0, help="MOG2 pixels per lane (tuning; 0 = default = 1)")
    args = ap.parse_args()

    pmc_child = bool(args.pmc_child)
    if pmc_child:
        args.main_only = True
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # --rehearse: the N > 1 control flow on ONE GPU (every rank on cuda:0, `gloo` instead of RCCL, the packed masks go through a
    # host copy): RCCL refuses two ranks on one device, and this box has one.  Numbers from it are not benchmark results.
    rehearse = args.rehearse and world > 1
    selftest = args.rccl_selftest and world == 1
    native = args.native_node and (world > 1 or selftest) and not rehearse
    if selftest:
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_PORT", "29517")):
            os.environ.setdefault(k, v)
    if world > 1 or selftest:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the process group comes up before this process makes its first GPU call
        if rehearse:
            dist.init_process_group("gloo")
            local = 0
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libbgs_hip has no CPU path")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from tracking_amd import Engine, capi
    from tracking_amd.sharding import MaskGather, packed_words, stream_block

    S = args.streams
    first_global, _ = stream_block(S * world, world, rank)
    # The host path FIRST, in a process that has not yet allocated and freed gigabytes of device memory: tools/r04_hostleg.py showed that
    # such a history (the copy calibration, a torch pool that came and went, a 7 GB engine) slows the multi-camera legs down 2-3x - 8
    # cameras with separately page-locked buffers 0.25 ms per frame in a fresh process, 0.57-0.84 ms after - while one synchronous camera
    # is unaffected; a property of the runtime's DMA path, not of the engine.  The same camera legs run AGAIN at the end of this process
    # (`host_path.after_large_allocations`) so that the line shows both.
    host = None
    if rank == 0 and world == 1 and not args.main_only and not selftest:
        hclip = make_source("sat", 1, dev, 1234).pool(25)[:, 0].cpu().numpy()
        host = host_leg(local, hclip)
        del hclip
    calibration = calibrate(local, S) if (rank == 0 and not pmc_child) else None  # before the model exists

    nd = None
    if native:
        from tracking_amd import node as bnode
        uid = [bnode.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(uid, src=0)
        nd = bnode.Node.rank(capi.MOG2, S * world, device=local, rank=rank, world=world, root_rank=0, uid=uid[0], flags=bnode.LOOPBACK if selftest else 0)
        nd.set_geometry(ROWS, COLS, CH)
        eng = Engine.from_handle(nd.engine_handle(0), capi.MOG2, S)
    else:
        eng = Engine(capi.MOG2, device=local, n_streams=S)
        eng.set_geometry(ROWS, COLS, CH)
    if args.px:
        eng.set_option(capi.OPT_MOG2_PIXELS_PER_LANE, args.px)
    fg = torch.empty((S, ROWS, COLS), dtype=torch.uint8, device=dev)
    bg = torch.empty((S, ROWS, COLS, CH), dtype=torch.uint8, device=dev) if args.with_bg else None
    words = packed_words(ROWS, COLS)
    gather = MaskGather(S, words, "cpu" if rehearse else dev, always_collective=selftest) if ((world > 1 or selftest) and not native) else None
    bits_dev = torch.empty((S, words), dtype=torch.int64, device=dev) if rehearse else None

    def step(frames):
        UPDATE_LAUNCHES[0] += 1
        if nd is not None:  # libbgs_node: kernel (packed masks into this step's gather buffer) + its share of the RCCL gather, all enqueued by the C++ driver
            nd.step_device([frames])
            return
        bits = bits_dev if rehearse else (gather.next_buffer() if gather else None)
        eng.process_batch_device(frames, fg, bg, bits)
        if rehearse:
            gather.next_buffer().copy_(bits_dev)  # synchronous D2H: rehearsal only
        if gather:
            gather.post()

    def drain():
        if gather:
            gather.drain()
        if nd is not None:
            nd.sync()
        torch.cuda.synchronize()

    # Set-up, untimed.  (1) saturation: the mixture model needs ~50 frames of S_sat before all K = 5 modes of every pixel are live.
    # (2) settle = AGEING the model: for its first ~100 frames the five weights of a pixel have not equalised yet, on every fifth
    # frame the matched mode is still the heaviest one, nothing is re-ordered and less is written (rounds 1-2 took this for a clock
    # burst; the round-2 verdict showed it is model age x write skipping: profiles/r02_mog2_launch_series.csv has a strict period of 5).
    # `value` must not depend on whether the driver asks for 20 or 2000 steps, so the model is aged here, on the same kernel, before
    # the W warm-up steps the contract asks for; the young model's first 20 launches are reported as `young_model_first_20`.
    # Frames: every frame the model ever sees carries FRESH noise (tools/synth.py SatStreams).  The untimed phases generate theirs one
    # by one into a scratch image; the W + K frames of the warm-up and the timed region are generated beforehand into a pool of DISTINCT
    # frames resident in HBM (the contract: inputs resident when the clock starts), capped at --pool-frames.
    SATURATE, SETTLE, SUSTAIN = 60, max(0, args.settle), max(0, args.sustain)
    src = make_source(args.input, S, dev, {"sat": 1234, "surv": 4321, "dense": 777}[args.input] + first_global)
    cur = torch.empty((S, ROWS, COLS, CH), dtype=torch.uint8, device=dev)
    for _ in range(SATURATE):
        step(src.into(cur))
    drain()
    eng.enable_kernel_timing(True)  # every launch from here on is timed (HIP events on the launch stream); slices below
    for _ in range(SETTLE):
        step(src.into(cur))
    drain()
    del cur
    period = min(max(25, (args.warmup + args.steps + 4) // 5 * 5), max(25, args.pool_frames // 5 * 5))  # a multiple of 5: S_sat's levels cycle with period 5
    pool = src.pool(period)
    ti = 0
    for _ in range(args.warmup):
        step(pool[ti % period])
        ti += 1
    drain()
    if world > 1 or selftest:
        dist.barrier()
    torch.cuda.synchronize()
    if gather:
        gather.reset_stats()
    if nd is not None:
        nd.step_stats(reset=True)
    mark_a = UPDATE_LAUNCHES[0]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(pool[ti % period])
        ti += 1
    drain()
    if world > 1 or selftest:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    marks = {"update": [mark_a, UPDATE_LAUNCHES[0]]}
    local_elapsed = elapsed
    if world > 1 or selftest:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if gather:
        gather_wait_ms = gather.blocked_s / max(1, args.steps) * 1e3  # of the timed region only (reset above)
    elif nd is not None:
        gather_wait_ms = nd.step_stats()[0] / max(1, args.steps)      # host ms per step inside bgs_node_step_device (it never blocks on the device)
    else:
        gather_wait_ms = 0.0
    # sustained leg (untimed by the contract's clock, reported beside it): SUSTAIN further launches of the same step
    clocks = ClockSampler(local) if rank == 0 else None
    if clocks:
        clocks.start()
    for _ in range(SUSTAIN):
        step(pool[ti % period])
        ti += 1
    drain()
    clocks = clocks.stop() if clocks else None
    selftest_ok = None
    if selftest and gather:  # what came through the RCCL gather is what the kernel wrote (an inverted frame: foreground everywhere, so the words are not all zero)
        bits = gather.next_buffer()
        eng.process_batch_device(255 - pool[ti % period], fg, bg, bits)
        UPDATE_LAUNCHES[0] += 1
        gather.post()
        got = gather.collect()
        torch.cuda.synchronize()
        selftest_ok = {"gather_equals_kernel_output": bool(torch.equal(got, gather.bufs[gather.last])), "nonzero_words": int(got.ne(0).sum().item()), "words": int(got.numel())}
    if selftest and nd is not None:  # the loop-back gather of the node driver against the same kernel's byte mask
        inv = 255 - pool[ti % period]
        nd.step_device([inv])
        UPDATE_LAUNCHES[0] += 1
        got = nd.copy_masks(torch.empty((S, words), dtype=torch.int64, device=dev))
        torch.cuda.synchronize()
        selftest_ok = {"transport": "libbgs_node, ncclSend to self + ncclRecv from self (the gathered words are held against the oracle in tests/test_gpu_07_node.py)",
                       "nonzero_words": int(got.ne(0).sum().item()), "words": int(got.numel())}
    _, _, k_name = eng.kernel_timing()
    series = eng.kernel_timing_series()
    eng.enable_kernel_timing(False)
    i0 = SETTLE + args.warmup
    timed_ms = series[i0:i0 + args.steps]
    k_ms, k_n = (float(timed_ms.mean()), int(timed_ms.size)) if timed_ms.size else (0.0, 0)
    # the library keeps at most 16384 per-launch timings: a very long run gets its kernel figures from a truncated slice (`value` is
    # wall-clock and unaffected); say so instead of silently averaging fewer launches
    timing_truncated = bool(timed_ms.size < args.steps)
    sus_ms = series[i0 + args.steps:i0 + args.steps + SUSTAIN]
    burst_ms = series[:20]
    if rank == 0 and args.series:
        with open(args.series, "w") as f:
            f.write("# per-launch duration of %s, HIP events on the launch stream; bench.py --steps %d --warmup %d --settle %d --sustain %d; "
                    "timing starts after %d untimed saturation launches\nlaunch,phase,ms\n" % (k_name, args.steps, args.warmup, SETTLE, SUSTAIN, SATURATE))
            for i, v in enumerate(series):
                phase = "settle" if i < SETTLE else "warmup" if i < i0 else "timed" if i < i0 + args.steps else "sustain"
                f.write("%d,%s,%.4f\n" % (i, phase, v))

    # N > 1 diagnostics (a sub-linear scaling curve must be explainable from the line alone): every rank's own kernel time over the
    # timed steps, its elapsed time, and the host time per step that the gather cost it
    per_rank = None
    if world > 1 or selftest:
        mine = torch.tensor([k_ms, gather_wait_ms, local_elapsed * 1e3 / max(1, args.steps)], dtype=torch.float64, device="cpu" if rehearse else dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        if rank == 0:
            per_rank = {"kernel_avg_ms": [round(float(v[0]), 4) for v in allr], "gather_wait_ms_per_step": [round(float(v[1]), 4) for v in allr],
                        "ms_per_step_local": [round(float(v[2]), 4) for v in allr],
                        "note": "one entry per rank: mean duration of the update kernel over the timed steps (HIP events); host time per step that the gather cost the rank "
                                "(torch path: blocked in MaskGather.next_buffer() waiting for the gather that last used the buffer; --native-node: time inside bgs_node_step_device, "
                                "which enqueues and never waits); the rank's own wall time per step; `ms_per_step` is the MAX over ranks"}
    live_modes = None
    single = None
    configs = None
    cpu = None
    surv = None
    dense = None
    clip = None
    probe = None
    clip0 = None
    if rank == 0:
        nm = eng.get_state("nmodes", (ROWS * COLS,), np.uint8, stream=0)
        live_modes = float(nm.mean())
        pr = eng.get_state("placement", (2,), np.float32)
        probe = {"chunk_MiB": int(pr[0]), "chunks": int(pr[1]),
                 "note": "the model is one virtual range backed by separately created physical chunks (hipMemCreate / hipMemMap; DESIGN.md 6.2; chunk_MiB 0 = one plain hipMalloc). "
                         "`dense_launch` = the placement witness: the same model streamed whole (BGS_OPT_MOG2_SPARSE = 0: every weight, summary, record and meta word read and "
                         "written back, 228 B/pixel) - its rate against calibration.copy_GBps_chunked says whether this model's placement is as good as a fresh chunked range's"}
        clip0 = pool[:25, 0].cpu().numpy()  # stream 0's frames for the host-path and CPU legs
    if rank == 0 and pmc_child:
        for T in (4, 8):
            clip_leg(eng, pool, period, S, T, launches=CLIP_PMC_LAUNCHES, warm=4)
    if rank == 0 and not args.main_only and not rehearse and nd is None:
        clip = {"note": "supplementary, never `value`: bgs_process_clip_device on the same engine and saturated model - T consecutive frames per launch, model kept in "
                        "registers across them, bit-identical results; for file-fed video or deployments that accept T-1 frame times of latency",
                "T4": clip_leg(eng, pool, period, S, 4), "T8": clip_leg(eng, pool, period, S, 8)}
    if rank == 0 and not args.main_only and not rehearse and nd is None and args.input == "sat":
        # placement witness: the whole model streamed (results unchanged: dense only writes back what it read)
        eng.set_option(capi.OPT_MOG2_SPARSE, 0)
        for i in range(3):
            eng.process_batch_device(pool[i % period], fg, None, None)
        torch.cuda.synchronize()
        eng.enable_kernel_timing(True)
        for i in range(10):
            eng.process_batch_device(pool[(3 + i) % period], fg, None, None)
        torch.cuda.synchronize()
        dms, _, _ = eng.kernel_timing()
        eng.enable_kernel_timing(False)
        eng.set_option(capi.OPT_MOG2_SPARSE, 3)
        dense_bytes = (2 * MODEL_BYTES_PER_PIXEL + 4) * S * ROWS * COLS
        probe["dense_launch"] = {"kernel_ms": round(dms, 4), "bytes_per_pixel": 2 * MODEL_BYTES_PER_PIXEL + 4, "GBps": round(dense_bytes / (dms * 1e-3) / 1e9, 1) if dms > 0 else None,
                                 "frac_of_box_copy": round(dense_bytes / (dms * 1e-3) / 1e9 / calibration["copy_GBps_chunked"], 4) if (dms > 0 and calibration and calibration.get("copy_GBps_chunked")) else None}
    if rank == 0 and not args.main_only:
        # BASELINE configs[1] literally: ONE 1080p stream.  Its 253 MB model fits the 256 MiB Infinity Cache, so this
        # number is not an HBM measurement; it is reported beside the batched one, never as `value`.
        e1 = Engine(capi.MOG2, device=local, n_streams=1)
        e1.set_geometry(ROWS, COLS, CH)
        fg1 = fg[:1]
        for i in range(30):
            e1.process_batch_device(pool[i % period, :1], fg1, None, None)
        torch.cuda.synchronize()
        e1.enable_kernel_timing(True)
        s0 = time.perf_counter()
        n1 = 200
        for i in range(n1):
            e1.process_batch_device(pool[i % period, :1], fg1, None, None)
        torch.cuda.synchronize()
        d1 = time.perf_counter() - s0
        ms1, _, _ = e1.kernel_timing()
        single = {"mpixels_per_s": round(n1 * ROWS * COLS / d1 / 1e6, 1), "kernel_ms": round(ms1, 4),
                  "kernel_GBps_moved": round(BYTES_PER_PIXEL * ROWS * COLS / (ms1 * 1e-3) / 1e9, 1),
                  "note": "single stream: 253 MB of model state fits the 256 MiB Infinity Cache (not an HBM figure)"}
        e1.close()
    if rank == 0 and (not args.main_only or pmc_child) and args.input == "sat":
        n_leg = PMC_LEG_LAUNCHES if pmc_child else 100
        del pool  # the scene legs bring their own frames
        pool = None
        torch.cuda.empty_cache()
        sv, spool = scene_leg(local, S, fg, "surv", steps=n_leg, sparse=3, marks=marks)
        surv = {"default": sv}
        if not pmc_child:
            surv["stores_only"], _ = scene_leg(local, S, fg, "surv", steps=n_leg, sparse=1, pool=spool, marks=marks)
        del spool
        torch.cuda.empty_cache()
        dense, dpool = scene_leg(local, S, fg, "dense", steps=n_leg, sparse=3, warm=200, marks=marks)
        del dpool
        torch.cuda.empty_cache()
    if pmc_child and rank == 0:
        with open(args.pmc_child, "w") as f:
            json.dump(marks, f)
    if rank == 0 and not args.main_only:
        if world == 1 and not selftest and host is not None:
            late = host_leg(local, clip0, cameras_only=True)
            late.pop("pcie_calibration", None)
            host["after_large_allocations"] = dict(late, note="the three 8-camera legs again at the END of this process, after the calibration ranges, the 7 GB model, 40 GB of frame pools "
                                                                "and the scene legs' engines have been allocated and freed: the runtime's multi-stream DMA path is slower then (tools/r04_hostleg.py)")
        if world == 1 and not args.no_configs and not selftest:
            # BASELINE configs[2] (WMV + ABL at 3840x2160) and configs[3] (SuBSENSE / LBSP at 1080p) in the driver-timed line, never `value`
            from tools import bench_configs
            eng.close()  # the 8.1 GB MOG2 model is not needed any more (its state was read above)
            eng = None
            try:
                configs = bench_configs.configs_block(device=local, cpu=not args.no_cpu_baseline)
            except Exception as ex:  # noqa: BLE001 - the headline must still be printed
                configs = {"error": repr(ex)}
        if not args.no_cpu_baseline and not rehearse:
            # N > 1: a shorter sample on rank 0 (the other ranks wait in the final barrier), so the line stays self-contained
            cpu = cpu_baseline(clip0, args.input, budget_s=16.0 if world == 1 else 6.0)

    if rank == 0 and not pmc_child:
        px_per_step_rank = S * ROWS * COLS
        total_px = px_per_step_rank * world * args.steps
        mpix = total_px / elapsed / 1e6
        algo_bytes = (BYTES_PER_PIXEL + (3 if args.with_bg else 0)) * px_per_step_rank
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic, traffic_source, traffic_detail = None, "none", None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if world == 1 and not args.main_only and not args.no_pmc and not args.with_bg and not selftest:
            pmc_all, why = pmc_traffic_live(S, args.steps, args.warmup, args.input, SETTLE)
            if pmc_all and "update" in pmc_all:
                traffic_detail = pmc_all["update"]
                traffic, traffic_source = traffic_detail["hbm_bytes_per_launch"], why
                for key, T in (("clip4", 4), ("clip8", 8)):  # the clip legs get their measured traffic too (never a byte model priced as bandwidth)
                    leg_c = clip and clip.get("T%d" % T)
                    if leg_c and key in pmc_all:
                        tb = pmc_all[key]
                        leg_c["traffic"] = tb
                        leg_c["traffic_B_per_pixel_frame"] = round(tb["hbm_bytes_per_launch"] / (px_per_step_rank * T), 2)
                        leg_c["hbm_GBps_measured"] = round(tb["hbm_bytes_per_launch"] / (leg_c["kernel_avg_ms"] * 1e-3) / 1e9, 1)
                for key, leg_s in (("surv", surv and surv.get("default")), ("dense", dense)):  # PMC traffic of the scene legs (the child's own, shorter, leg on the same scene)
                    if leg_s and key in pmc_all:
                        tb = pmc_all[key]
                        leg_s["traffic"] = tb
                        leg_s["traffic_B_per_pixel"] = round(tb["hbm_bytes_per_launch"] / px_per_step_rank, 2)
                        leg_s["hbm_GBps_measured"] = round(tb["hbm_bytes_per_launch"] / (leg_s["kernel_ms"] * 1e-3) / 1e9, 1)
                        leg_s["hbm_frac_of_peak_measured"] = round(tb["hbm_bytes_per_launch"] / (leg_s["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
            else:
                traffic_source = "live PMC passes unavailable (%s); " % why
        if traffic is None and os.path.exists(pmc) and S == 32 and args.input == "sat" and not args.with_bg:  # (the constants are for the default workload)
            try:
                consts = json.load(open(pmc))
                rec = consts.get("mog2_update_kernel", {})
                traffic = rec.get("hbm_bytes_per_launch")
                traffic_detail = {k: rec[k] for k in ("hbm_bytes_per_launch", "read_bytes", "write_bytes") if k in rec}
                traffic_source = (traffic_source if traffic_source != "none" else "") + "NOT measured in this run: constant read from profiles/pmc_traffic.json (%s)" % rec.get("source", "rocprofv3 --pmc passes of the same workload")
                for key, leg_s in (("s_surv", surv and surv.get("default")), ("s_dense", dense)):  # the scene legs fall back the same way, and say so
                    if leg_s and "traffic" not in leg_s and key in consts:
                        tb = {k: consts[key][k] for k in ("hbm_bytes_per_launch", "read_bytes", "write_bytes")}
                        leg_s["traffic"] = tb
                        leg_s["traffic_source"] = "NOT measured in this run: constant read from profiles/pmc_traffic.json (%s)" % consts[key].get("source", "")
                        leg_s["traffic_B_per_pixel"] = round(tb["hbm_bytes_per_launch"] / px_per_step_rank, 2)
                        leg_s["hbm_GBps_measured"] = round(tb["hbm_bytes_per_launch"] / (leg_s["kernel_ms"] * 1e-3) / 1e9, 1)
            except Exception:
                traffic = None

        def leg(ms):
            if not len(ms):
                return None
            a = algo_bytes / (float(ms.mean()) * 1e-3) / 1e9
            return {"launches": int(len(ms)), "kernel_avg_ms": round(float(ms.mean()), 4), "kernel_min_ms": round(float(ms.min()), 4), "kernel_max_ms": round(float(ms.max()), 4),
                    "achieved": round(a, 1), "frac": round(a / HBM_PEAK_GBPS, 4)}
        box_copy = calibration.get("copy_GBps_chunked") if calibration else None
        in_name = {"sat": "S_sat", "surv": "S_surv", "dense": "S_dense"}[args.input]
        if world > 1:
            gather_desc = ("REHEARSAL: gloo gather through host copies, all ranks on one GPU - not a benchmark" if rehearse else
                           "libbgs_node (include/bgs_node.h): root posts one ncclRecv per peer, every peer one ncclSend, grouped, on a second HIP stream, double-buffered" if native else
                           "RCCL gather of bit-packed masks to rank 0 every step (torch.distributed.gather), overlapped")
        else:
            gather_desc = (("RCCL SELF-TEST through libbgs_node: one rank, its block sent to itself and received from itself every step" if native else
                            "RCCL SELF-TEST: one rank, gather / barrier / all-reduce issued anyway (the N > 1 code path on one GPU; not a scaling result)") if selftest else "none (1 GPU)")
        out = {
            "metric": "MixtureOfGaussianV2BGS throughput (Mpixels/s; concurrent 1080p30 streams in streams_1080p30)",
            "value": round(mpix, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "MixtureOfGaussianV2BGS (MOG2 K=5, alpha=0.05, threshold 15) on 1920x1080x3 uint8, %s synthetic input, %d streams per GPU batched in one launch "
                                   "(BASELINE configs[1] geometry x %d = the per-GPU share of configs[4]); frames resident in HBM" % (in_name, S, S),
                       "streams_per_gpu": S, "rows": ROWS, "cols": COLS, "channels": CH, "K": 5, "input": args.input,
                       "frames": "fresh noise in every frame the model sees (tools/synth.py); %d distinct frames resident in HBM for the %d warm-up + %d timed steps%s" % (
                           period, args.warmup, args.steps, "" if period >= args.warmup + args.steps else " (cycled: --pool-frames %d)" % args.pool_frames),
                       "mask_gather": gather_desc},
            "streams_1080p30": round(mpix / (ROWS * COLS / 1e6) / 30.0, 1),
            "frames_per_s": round(mpix * 1e6 / (ROWS * COLS), 1),
            "mean_live_modes_stream0": live_modes,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": traffic, "traffic_source": traffic_source, "traffic_read_write": traffic_detail, "kernel": k_name, "kernel_avg_ms": round(k_ms, 4), "kernel_launches": k_n, "kernel_timing_truncated": timing_truncated,
                         "algorithmic_bytes_per_launch": algo_bytes, "bytes_per_pixel": BYTES_PER_PIXEL + (3 if args.with_bg else 0),
                         "bytes_per_pixel_derivation": "r 3 frame + 2 meta + 20 weights + 10 summaries + 16 the one record the summaries cannot rule out; w 20 weights + 16 that record + 2 meta + 1 mask (+2 when its summary is rewritten) (DESIGN.md 6.1)",
                         "frac_of_achievable_6290": round(achieved / 6290.0, 4),
                         "frac_of_box_copy": round(achieved / box_copy, 4) if box_copy else None,
                         "frac_of_box_copy_note": "achieved / calibration.copy_GBps_chunked: the same kernel time against what a float4 copy reaches on THIS box through the model's own allocation scheme",
                         "vs_survey_206B": {"note": "the same kernel time priced at SURVEY.md 8(d)'s 206 B/pixel (the reference's sorted-array formulation): an EQUIVALENT rate, comparable with rounds 1-2, not bytes moved - it may exceed the peak",
                                            "equivalent_GBps": round(SURVEY_BYTES_PER_PIXEL * px_per_step_rank / (k_ms * 1e-3) / 1e9, 1) if k_ms > 0 else 0.0,
                                            "equivalent_frac_of_peak": round(SURVEY_BYTES_PER_PIXEL * px_per_step_rank / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if k_ms > 0 else 0.0},
                         "timed_region": "the K timed steps, after %d saturation + %d model-ageing (settle) + %d warm-up launches: steady-state traffic" % (SATURATE, SETTLE, args.warmup),
                         "sustained": leg(sus_ms), "young_model_first_20": leg(burst_ms)},
            "cpu_baseline": cpu,
            "calibration": dict(calibration, clocks=clocks) if calibration else {"clocks": clocks},
            "per_rank": per_rank,
            "model_placement": probe,
            "rccl_selftest_gather_matches_kernel_output": selftest_ok,
            "single_stream": single,
            "host_path": host,
            "s_surv": surv,
            "s_dense": dense,
            "clip": clip,
            "configs": configs,
        }
        print(json.dumps(out), flush=True)
    if eng is not None:
        eng.close()
    if nd is not None:
        nd.close()
    if world > 1 or selftest:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
