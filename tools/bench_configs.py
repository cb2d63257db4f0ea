#!/usr/bin/env python3
"""Secondary workloads of BASELINE.json (configs[2], configs[3]) and the other byte kernels, kernel time by HIP events.
Not the driver's bench line (bench.py is); the numbers go to DESIGN.md §6.
  configs[2]: WeightedMovingVarianceBGS + AdaptiveBackgroundLearning, 3840x2160 (HBM-bound stress)
  configs[3]: LBSP descriptor path, 1920x1080
usage: bench_configs.py [--streams S] [--swizzle 0|1]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tools import synth
from tracking_amd import Engine, capi
from tracking_amd.engine import lbsp_describe_device


def cpu_rate(algo, frames_np, warm=3, params=None):
    """The oracle (CPU restatement, 1 thread; rows split over the box's quota for the MOG loops) on a few frames of the same size."""
    from oracle import pyoracle
    import os
    threads = 1
    if algo in (capi.MOG2, capi.MOG1):
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            threads = min(len(os.sched_getaffinity(0)), int(int(q) / int(per))) if q != "max" else 16
        except Exception:
            threads = 16
    o = pyoracle.Oracle(algo, params=params, threads=threads)
    for f in frames_np[:warm]:
        o.process(f, want_bg=False)
    t0 = time.perf_counter()
    n = 0
    for f in frames_np[warm:]:
        o.process(f, want_bg=False)
        n += 1
    dt = time.perf_counter() - t0
    return n * frames_np.shape[1] * frames_np.shape[2] / dt / 1e6, threads


# Bytes MOVED per pixel by the kernels whose traffic depends on the data (the mixture models read only the modes a pixel has and write
# only what changed; GMG walks a per-pixel list): measured with rocprofv3 --pmc (FETCH_SIZE x 2 + WRITE_SIZE) on these very legs -
# profiles/r03_mog1_pmc.txt, profiles/r03_dp_pmc.txt, DESIGN.md 9 (GMG).  Round 3 priced these legs at the DENSE sorted-array figure of
# SURVEY.md 8(a) and printed "fractions" of 150-240 %: not fractions of anything.  A leg without a measurement prints no fraction.
MOVED_BPP = {(capi.MOG1, "surv"): 97, (capi.MOG1, "sat"): 171, (capi.DP_ZIVKOVIC_AGMM, "sat"): 98, (capi.DP_ZIVKOVIC_AGMM, "surv"): 42,
             (capi.DP_GRIMSON_GMM, "sat"): 122, (capi.GMG, "surv"): 76}
DATA_DEPENDENT = (capi.MOG1, capi.MOG2, capi.DP_ZIVKOVIC_AGMM, capi.DP_GRIMSON_GMM, capi.GMG)


def run(algo, name, rows, cols, S, bpp, steps=60, borrow=True, want_bg=False, cpu_frames=6, cpu_warm=3, params=None, kind="surv"):
    dev = torch.device("cuda", 0)
    T = 10 if kind == "sat" else 8  # S_sat repeats every 5 frames: the pool must wrap at a multiple of 5
    pool = torch.empty((T, S, rows, cols, 3), dtype=torch.uint8, device=dev)
    for s in range(S):
        pool[:, s] = (synth.s_surv if kind == "surv" else synth.s_sat)(T, rows, cols, seed=4321 + s, device=dev)
    e = Engine(algo, n_streams=S, params=params)
    e.set_geometry(rows, cols, 3)
    if borrow:
        e.set_option(capi.OPT_BORROW_FRAMES, 1)
    fg = torch.empty((S, rows, cols), dtype=torch.uint8, device=dev)
    bg = torch.empty((S, rows, cols, 3), dtype=torch.uint8, device=dev) if want_bg else None
    # mixture models on S_sat need ~100 frames before the weights have equalised and every frame re-orders the modes (steady-state
    # traffic; a younger model writes less and flatters the figure - round-2 verdict); everything else is warm after 10
    mixture = algo in (capi.MOG2, capi.MOG1, capi.DP_ZIVKOVIC_AGMM, capi.DP_GRIMSON_GMM)
    nwarm = 140 if (mixture and kind == "sat") else (40 if mixture else 10)
    for t in range(nwarm):
        e.process_batch_device(pool[t % T], fg, bg, None)
    torch.cuda.synchronize()
    e.enable_kernel_timing(True)
    t0 = time.perf_counter()
    for t in range(steps):
        e.process_batch_device(pool[(nwarm + t) % T], fg, bg, None)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms, n, kname = e.kernel_timing()
    px = S * rows * cols
    cpu = ""
    if cpu_frames:
        sample = pool[:cpu_frames, 0].cpu().numpy() if T >= cpu_frames else torch.cat([pool[:, 0]] * (cpu_frames // T + 1))[:cpu_frames].cpu().numpy()
        rate, th = cpu_rate(algo, sample, warm=cpu_warm, params=params)
        cpu = " | CPU oracle %.1f Mpix/s (%d thread%s)" % (rate, th, "s" if th > 1 else "")
    if algo in DATA_DEPENDENT:
        moved = MOVED_BPP.get((algo, kind))
        rate = ("%7.1f GB/s moved (%d B/px by PMC on this leg) = %.1f%% of 8 TB/s" % (moved * px / ms / 1e6, moved, moved * px / ms / 1e6 / 80.0)) if moved else "traffic data-dependent, not measured on this leg: no fraction"
        rate += " [dense sorted-array figure of SURVEY.md 8(a): %d B/px - not moved]" % bpp
    else:
        rate = "%7.1f GB/s algorithmic (%d B/px) = %.1f%% of 8 TB/s" % (bpp * px / ms / 1e6, bpp, bpp * px / ms / 1e6 / 80.0)
    print("%-34s %dx%d x%d streams: kernel %-18s %.4f ms  -> %8.1f Mpix/s  %s | wall %.1f Mpix/s%s"
          % (name, cols, rows, S, kname, ms, px / ms / 1e3, rate, px * steps / wall / 1e6, cpu))
    e.close()


def run_subsense(S, steps=30, kind="surv", algo=None, label="SuBSENSEBGS", warm=6, groups=1):
    """BASELINE configs[3]: SuBSENSE at 1920x1080 (per-frame wall time: ~20 launches incl. the flood-fill host loop)."""
    dev = torch.device("cuda", 0)
    rows, cols, T = 1080, 1920, 8
    pool = torch.empty((T, S, rows, cols, 3), dtype=torch.uint8, device=dev)
    for s in range(S):
        pool[:, s] = (synth.s_surv if kind == "surv" else synth.s_smooth)(T, rows, cols, seed=4321 + s, device=dev)
    e = Engine(capi.SUBSENSE if algo is None else algo, n_streams=S)
    e.set_geometry(rows, cols, 3)
    fg = torch.empty((S, rows, cols), dtype=torch.uint8, device=dev)
    # groups > 1: the batch as `groups` stream ranges, each on its own HIP stream (bgs_process_range_device): one range's memory-bound
    # tail (sample writes, post-processing) then runs beside another range's VALU-bound phase A
    hs = [torch.cuda.Stream() for _ in range(groups)] if groups > 1 else None
    per = S // groups

    def step(t):
        if groups == 1:
            e.process_batch_device(pool[t % T], fg, None, None)
        else:
            for g in range(groups):
                e.process_batch_device(pool[t % T, g * per:(g + 1) * per], fg[g * per:(g + 1) * per], None, None, hip_stream=hs[g].cuda_stream, first=g * per, count=per)
    for t in range(warm):
        step(t)
    torch.cuda.synchronize()
    e.enable_kernel_timing(True)
    t0 = time.perf_counter()
    for t in range(steps):
        step(warm + t)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    ms, n, kname = e.kernel_timing()
    px = S * rows * cols
    print("%-34s %dx%d x%d streams: %.3f ms/frame-step wall -> %8.1f Mpix/s (%.1f 1080p frames/s); %s %.3f ms; fg ratio %.3f"
          % ("%s (%s input%s%s)" % (label, kind, "" if warm == 6 else ", model aged %d frames" % warm, "" if groups == 1 else ", %d ranges on %d HIP streams" % (groups, groups)), cols, rows, S, wall * 1e3, px / wall / 1e6, S / wall, kname, ms, float((fg != 0).float().mean())))
    e.close()


def run_lbsp():
    # LBSP descriptors, 1080p
    img = synth.s_surv(1, 1080, 1920, seed=9, device="cuda")[0]
    from oracle import pyoracle
    lut = pyoracle.lbsp_lut(0.333, 0, 3)
    out = torch.empty((1080, 1920, 3), dtype=torch.int16, device="cuda")
    for _ in range(5):
        lbsp_describe_device(img, lut, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        lbsp_describe_device(img, lut, out=out)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 50
    px = 1080 * 1920
    print("%-34s 1920x1080 x1: %.4f ms -> %8.1f Mpix/s  %7.1f GB/s algorithmic (9 B/px)" % ("LBSP descriptors (lbsp_kernel)", ms, px / ms / 1e3, 9 * px / ms / 1e6))
    from tracking_amd.engine import lbsp_describe_batch_device
    imgs = synth.s_surv(16, 1080, 1920, seed=9, device="cuda")
    for _ in range(3):
        lbsp_describe_batch_device(imgs, lut)
    torch.cuda.synchronize()
    a.record()
    for _ in range(20):
        lbsp_describe_batch_device(imgs, lut)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    px = 16 * 1080 * 1920
    print("%-34s 1920x1080 x16 (one launch): %.4f ms -> %8.1f Mpix/s  %7.1f GB/s algorithmic (9 B/px, incl. the output allocation)" % ("LBSP descriptors (lbsp_kernel)", ms, px / ms / 1e3, 9 * px / ms / 1e6))



def run_pipeline(S=8, steps=30):
    """Frames in HBM -> SuBSENSE masks -> blob rectangles, all on the device; only the boxes and offsets cross PCIe."""
    from tracking_amd.engine import mask_components_batch_device
    dev = torch.device("cuda", 0)
    rows, cols, T = 1080, 1920, 8
    pool = torch.empty((T, S, rows, cols, 3), dtype=torch.uint8, device=dev)
    for s in range(S):
        pool[:, s] = synth.s_surv(T, rows, cols, seed=4321 + s, device=dev)
    e = Engine(capi.SUBSENSE, n_streams=S)
    e.set_geometry(rows, cols, 3)
    fg = torch.empty((S, rows, cols), dtype=torch.uint8, device=dev)
    for t in range(6):
        e.process_batch_device(pool[t % T], fg, None, None)
    torch.cuda.synchronize()
    nbox = nbytes = 0
    t0 = time.perf_counter()
    for t in range(steps):
        e.process_batch_device(pool[(6 + t) % T], fg, None, None)
        _, boxes, off = mask_components_batch_device(fg, 8, max_boxes=65536)
        host = boxes.cpu()
        nbox += host.shape[0]
        nbytes += host.numel() * 4 + off.numel() * 4
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    print("pipeline SuBSENSE -> components  1920x1080 x%d streams: %.3f ms per step = %.1f 1080p frames/s end to end; %.1f boxes and %.0f bytes D2H per step "
          "(full masks would be %d bytes)" % (S, wall * 1e3, S / wall, nbox / steps, nbytes / steps, S * rows * cols))
    # The same, as a streaming consumer would drive it: nothing waits for step t before step t+1 is enqueued - two sets of output
    # buffers, the boxes of step t go to pinned memory asynchronously and are read when the event behind that copy has fired, one step late.
    import ctypes as C
    max_boxes = 65536
    lib = capi.lib()
    wsz = lib.bgs_mask_components_batch_workspace(S, rows, cols)
    fgs = [torch.empty((S, rows, cols), dtype=torch.uint8, device=dev) for _ in range(2)]
    boxes = [torch.zeros((max_boxes, 6), dtype=torch.int32, device=dev) for _ in range(2)]
    offs = [torch.zeros(S + 1, dtype=torch.int32, device=dev) for _ in range(2)]
    work = [torch.empty(wsz, dtype=torch.uint8, device=dev) for _ in range(2)]
    hbox = [torch.zeros((4096, 6), dtype=torch.int32).pin_memory() for _ in range(2)]
    hoff = [torch.zeros(S + 1, dtype=torch.int32).pin_memory() for _ in range(2)]
    ev = [torch.cuda.Event() for _ in range(2)]
    stream = torch.cuda.current_stream().cuda_stream
    nbox = 0
    e.close()
    e = Engine(capi.SUBSENSE, n_streams=S)  # a fresh model, the same 6 warm-up frames: comparable with the figure above
    e.set_geometry(rows, cols, 3)
    for t in range(6):
        e.process_batch_device(pool[t % T], fg, None, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(steps + 1):
        i = t & 1
        if t < steps:
            e.process_batch_device(pool[(6 + t) % T], fgs[i], None, None)
            capi.check(lib.bgs_mask_components_batch_device(0, C.c_void_p(fgs[i].data_ptr()), S, rows, cols, 8, None, C.c_void_p(boxes[i].data_ptr()), max_boxes,
                                                            C.c_void_p(offs[i].data_ptr()), C.c_void_p(work[i].data_ptr()), C.c_void_p(stream)))
            hbox[i].copy_(boxes[i][:4096], non_blocking=True)
            hoff[i].copy_(offs[i], non_blocking=True)
            ev[i].record()
        if t >= 1:
            j = (t - 1) & 1
            ev[j].synchronize()
            nbox += int(hoff[j][-1])
    wall2 = (time.perf_counter() - t0) / steps
    print("pipeline, consumer one step behind   1920x1080 x%d streams: %.3f ms per step = %.1f 1080p frames/s end to end; %.1f boxes per step"
          % (S, wall2 * 1e3, S / wall2, nbox / steps))
    e.close()


def run_dp():
    """N4: the package_bgs/dp models at 1080p x 16 streams (state r/w + frame + mask bytes per pixel)."""
    run(capi.DP_ZIVKOVIC_AGMM, "DPZivkovicAGMMBGS (K=3, S_sat)", 1080, 1920, 16, 126, borrow=False, kind="sat", cpu_frames=0)
    run(capi.DP_GRIMSON_GMM, "DPGrimsonGMMBGS (K=3, S_sat)", 1080, 1920, 16, 150, borrow=False, kind="sat", cpu_frames=0)
    run(capi.DP_ZIVKOVIC_AGMM, "DPZivkovicAGMMBGS (K=3, S_surv)", 1080, 1920, 16, 126, borrow=False)
    run(capi.DP_GRIMSON_GMM, "DPGrimsonGMMBGS (K=3, S_surv)", 1080, 1920, 16, 150, borrow=False)
    run(capi.DP_WREN_GA, "DPWrenGABGS", 1080, 1920, 16, 36, borrow=False)
    run(capi.DP_MEAN, "DPMeanBGS", 1080, 1920, 16, 28, borrow=False)
    run(capi.DP_ADAPTIVE_MEDIAN, "DPAdaptiveMedianBGS", 1080, 1920, 16, 7, borrow=False)  # 3 frame + 3 median + 1 mask (+3/7 write-back); state is MALL-resident


def run_cc():
    """N1: connected components of a 1080p foreground-like mask (blobs + salt noise) and of a worst case (random 45 %)."""
    import numpy as np
    from tracking_amd.engine import mask_components_device
    rng = np.random.default_rng(1)
    blobs = np.zeros((1080, 1920), np.uint8)
    for _ in range(60):
        y, x = rng.integers(0, 1000), rng.integers(0, 1800)
        blobs[y:y + rng.integers(5, 80), x:x + rng.integers(5, 120)] = 255
    blobs[rng.random(blobs.shape) < 0.001] = 255
    for name, m in (("blobs+noise", blobs), ("random 45%", np.where(rng.random((1080, 1920)) < 0.45, 255, 0).astype(np.uint8)), ("full", np.full((1080, 1920), 255, np.uint8))):
        d = torch.from_numpy(m).cuda()
        for _ in range(3):
            labels, boxes, n = mask_components_device(d, 8, max_boxes=65536)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            labels, boxes, n = mask_components_device(d, 8, max_boxes=65536)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        print("connected components 1920x1080 %-12s: %d components, %.3f ms per mask (incl. count read-back) -> %.1f Mpix/s" % (name, n, ms, 1080 * 1920 / ms / 1e3))


def run_clip(S=32, rows=1080, cols=1920, kind="sat", steps=48, Ts=(1, 2, 4, 8), algo=None, dense=201.0, label="MOG2"):
    """bgs_process_clip_device on the bench geometry: T frames of every stream per launch, the model held in registers."""
    dev = torch.device("cuda", 0)
    P = 16  # a pool of 16 time steps; S_sat has period 5, so a clip may start at any multiple of 5... the pool is walked cyclically in whole clips
    pool = torch.empty((P, S, rows, cols, 3), dtype=torch.uint8, device=dev)
    gen = synth.s_sat if kind == "sat" else synth.s_surv
    for s in range(S):
        pool[:, s] = gen(P, rows, cols, seed=4321 + s, device=dev)
    px = S * rows * cols
    for T in Ts:
        e = Engine(algo if algo is not None else capi.MOG2, n_streams=S)
        e.set_geometry(rows, cols, 3)
        bits = torch.empty((T, S, rows * cols // 64), dtype=torch.int64, device=dev)
        for t in range(0, 160, T):  # saturate AND age the mixture (every mode of every pixel live on S_sat, weights equalised: steady-state traffic)
            e.process_clip_device(pool[(t % P):(t % P) + T], T, None, None, bits)
        torch.cuda.synchronize()
        for i in range(12):  # auto mode reads its samples after the sync above and may switch kernels (a switch to the filter kernel rebuilds the summaries once): not part of the timing
            t = (160 + i * T) % P
            e.process_clip_device(pool[t:t + T], T, None, None, bits)
        torch.cuda.synchronize()
        e.enable_kernel_timing(True)
        t0 = time.perf_counter()
        for i in range(steps):
            t = (160 + i * T) % P
            e.process_clip_device(pool[t:t + T], T, None, None, bits)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        ms, n, kname = e.kernel_timing()
        moved = dense / T + 3 + 1 / 8.0
        fused = "clip" in kname or T == 1 or kname == "dp_gmm_kernel"  # one launch per clip call; otherwise T launches, and only the wall clock says what a frame costs
        fms = ms / T if fused else wall * 1e3 / (steps * T)
        print(label + " clip T=%d (%s) %dx%d x%d streams: %-18s %.3f ms per launch, %.3f ms per frame step -> %7.1f Gpix/s = %6.0f 1080p frames/s; "
              "%d B/px/frame algorithmic -> %.2f of 8 TB/s; dense model bytes moved %.1f B/px/frame; wall %.1f Gpix/s"
              % (T, kind, cols, rows, S, kname, ms, fms, px / fms / 1e6, px / fms * 1e3 / (rows * cols), dense + 5, (dense + 5.0) * px / fms / 1e9 / 8.0, moved,
                 px * T * steps / wall / 1e9))
        e.close()


HBM_PEAK_GBPS = 8000.0


def _leg(ms, px, bpp):
    gbps = bpp * px / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    return {"kernel_avg_ms": round(ms, 4), "bytes_per_pixel": bpp, "mpixels_per_s": round(px / ms / 1e3, 1) if ms > 0 else 0.0,
            "achieved_GBps": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBPS, 4)}


def _cpu_leg(algo, frames_np, warm, label):
    rate, th = cpu_rate(algo, frames_np, warm=warm)
    return {"value": round(rate, 2), "unit": "Mpixels/s", "cores": th, "kind": "port",
            "sample": "%d timed frames %dx%d of stream 0 (same synthetic clip, %d warm-up frames), oracle %s" % (len(frames_np) - warm, frames_np.shape[2], frames_np.shape[1], warm, label)}


def configs_block(device=0, S=8, steps=40, cpu=True):
    """BASELINE configs[2] and configs[3] for bench.py's JSON line (never `value`): kernel time by HIP events on the launch stream,
    algorithmic bytes per pixel of SURVEY.md 8(a), fraction of the 8 TB/s peak, the CPU oracle on a bounded sample beside each."""
    dev = torch.device("cuda", device)
    out = {"note": "BASELINE configs[2] and configs[3], supplementary - never `value`; kernel time = HIP events on the launch stream, bytes per pixel = SURVEY.md 8(a) "
                   "(algorithmic, data-independent where the path is pointwise), frac = of the 8 TB/s HBM peak; cpu = the oracle (port) on a bounded sample of the same frames"}
    # ---- configs[2]: WeightedMovingVarianceBGS + AdaptiveBackgroundLearning, 3840x2160, back to back on the same frames
    rows, cols, T = 2160, 3840, 8
    pool = torch.empty((T, S, rows, cols, 3), dtype=torch.uint8, device=dev)
    for s in range(S):
        pool[:, s] = synth.s_surv(T, rows, cols, seed=4321 + s, device=dev)
    wmv, abl = Engine(capi.WMV, device=device, n_streams=S), Engine(capi.ABL, device=device, n_streams=S)
    for e in (wmv, abl):
        e.set_geometry(rows, cols, 3)
    wmv.set_option(capi.OPT_BORROW_FRAMES, 1)  # the caller's previous frames ARE the history: the algorithmic 10 B/px (the default copies each frame into a private ring)
    fg1 = torch.empty((S, rows, cols), dtype=torch.uint8, device=dev)
    fg2 = torch.empty((S, rows, cols), dtype=torch.uint8, device=dev)

    def step23(t):  # (the background image of ABL IS its uint8 state - SURVEY.md 8a a5 counts it once; no separate copy is asked for)
        wmv.process_batch_device(pool[t % T], fg1, None, None)
        abl.process_batch_device(pool[t % T], fg2, None, None)
    for t in range(10):
        step23(t)
    torch.cuda.synchronize()
    wmv.enable_kernel_timing(True), abl.enable_kernel_timing(True)
    t0 = time.perf_counter()
    for t in range(steps):
        step23(10 + t)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    px = S * rows * cols
    ms_w, _, k_w = wmv.kernel_timing()
    ms_a, _, k_a = abl.kernel_timing()
    c2 = {"workload": "WeightedMovingVarianceBGS + AdaptiveBackgroundLearning back to back on the same frames, %d x 3840x2160x3 uint8 S_surv in HBM, one launch per class and step" % S,
          "wmv": dict(_leg(ms_w, px, 10), kernel=k_w), "abl": dict(_leg(ms_a, px, 10), kernel=k_a),
          "both_kernels": _leg(ms_w + ms_a, px, 20), "ms_per_step_wall": round(wall, 4), "frames_4k_per_s": round(S / (wall * 1e-3), 1)}
    if cpu:
        sample = pool[:6, 0].cpu().numpy()
        c2["wmv"]["cpu_baseline"] = _cpu_leg(capi.WMV, sample, 3, "WeightedMovingVarianceBGS, 1 thread")
        c2["abl"]["cpu_baseline"] = _cpu_leg(capi.ABL, sample, 3, "AdaptiveBackgroundLearning, 1 thread")
    wmv.close(), abl.close()
    # the same two classes as ONE fused launch (bgs_group, kernel_fanout.h): one read of the frame and of the shared history
    from tracking_amd.engine import Group
    grp = Group([capi.WMV, capi.ABL], device=device, n_streams=S)
    grp.set_geometry(rows, cols, 3)
    grp.set_option(capi.OPT_BORROW_FRAMES, 1)
    for t in range(10):
        grp.process_batch_device(pool[t % T], [fg1, fg2], None)
    torch.cuda.synchronize()
    grp.enable_kernel_timing(True)
    t0 = time.perf_counter()
    for t in range(steps):
        grp.process_batch_device(pool[(10 + t) % T], [fg1, fg2], None)
    torch.cuda.synchronize()
    wall_g = (time.perf_counter() - t0) / steps * 1e3
    ms_g, _ = grp.kernel_timing()
    # r 3 x 3 (frame, t-1, t-2) + 3 (ABL background) ; w 3 (ABL background) + 1 + 1 (masks) = 17 B/pixel instead of 10 + 10
    c2["fused_group"] = dict(_leg(ms_g, px, 17), kernel="fan_kernel", ms_per_step_wall=round(wall_g, 4), frames_4k_per_s=round(S / (wall_g * 1e-3), 1),
                             speedup_vs_both_kernels=round((ms_w + ms_a) / ms_g, 3) if ms_g > 0 else None,
                             note="bgs_group of the two classes (kernel_fanout.h): ONE launch per step over one read of the frame and of the shared history - "
                                  "17 B/pixel moved (3 frames + ABL state in and out + 2 masks) instead of 10 + 10; outputs bit-identical to the two engines")
    grp.close()
    del pool, fg1, fg2
    out["configs2_wmv_abl_4k"] = c2
    # ---- configs[3]: SuBSENSE (LBSP descriptor path) at 1920x1080: whole frame step, young and aged model; lbsp_kernel alone
    # frames: fresh sensor noise in every frame the model sees (tools/synth.py SurvStreams; round 3 cycled a pool of 8 frames, whose values
    # the 50-sample model then holds exactly): the timed steps run over pools of DISTINCT frames resident in HBM, the untimed ageing
    # generates its frames one by one
    rows, cols, T = 1080, 1920, 36
    src = synth.SurvStreams(S, rows, cols, seed0=4321, device=dev)
    pool = src.pool(T)
    cur = torch.empty((S, rows, cols, 3), dtype=torch.uint8, device=dev)
    e = Engine(capi.SUBSENSE, device=device, n_streams=S)
    t0 = time.perf_counter()
    e.set_geometry(rows, cols, 3)
    fg = torch.empty((S, rows, cols), dtype=torch.uint8, device=dev)
    e.process_batch_device(pool[0], fg, None, None)  # the first frame initialises the model (refreshModel(1.0): 50 samples per pixel)
    torch.cuda.synchronize()
    init_ms = (time.perf_counter() - t0) * 1e3
    px = S * rows * cols
    c3 = {"workload": "SuBSENSEBGS (LBSP + colour sample consensus, 50 samples per pixel, feedback, post-processing), %d x 1920x1080x3 uint8 S_surv in HBM (fresh sensor noise in every frame), all streams per launch" % S,
          "bytes_per_pixel_note": "data-dependent; SURVEY.md 8(a) a11 gives >= 110 B/pixel (two matching samples, the float maps, frame and mask): `frac` prices the whole step at that floor",
          "first_frame_ms_incl_allocation": round(init_ms, 2)}
    t_seen = 1

    two = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)] if S >= 2 else None

    def step_two_ranges(frames):  # the same batch as two stream ranges, each on a HIP stream of its own (bgs_process_range_device)
        h = S // 2
        e.process_batch_device(frames[:h], fg[:h], None, None, hip_stream=two[0].cuda_stream, first=0, count=h)
        e.process_batch_device(frames[h:], fg[h:], None, None, hip_stream=two[1].cuda_stream, first=h, count=S - h)

    def timed_two_ranges(n):
        """Not a different workload: the S cameras driven as two ranges on two HIP streams, as a host with several capture threads would -
        one range's phase B and post-processing then run beside the other's phase A once the two have drifted apart."""
        nonlocal t_seen
        if two is None:
            return None
        torch.cuda.synchronize()
        for _ in range(3):
            step_two_ranges(pool[t_seen % T])
            t_seen += 1
        torch.cuda.synchronize()
        w0 = time.perf_counter()
        for _ in range(n):
            step_two_ranges(pool[t_seen % T])
            t_seen += 1
        torch.cuda.synchronize()
        wall_ms = (time.perf_counter() - w0) / n * 1e3
        return {"ms_per_step_wall": round(wall_ms, 4), "frames_1080p_per_s": round(S / (wall_ms * 1e-3), 1),
                "note": "the same %d cameras and frames as ranges [0, %d) and [%d, %d) on two HIP streams, %d steps" % (S, S // 2, S // 2, S, n)}

    def timed(n):
        nonlocal t_seen
        e.enable_kernel_timing(True)
        torch.cuda.synchronize()
        w0 = time.perf_counter()
        for _ in range(n):
            e.process_batch_device(pool[t_seen % T], fg, None, None)
            t_seen += 1
        torch.cuda.synchronize()
        wall_ms = (time.perf_counter() - w0) / n * 1e3
        ms, _, kname = e.kernel_timing()
        e.enable_kernel_timing(False)
        # the same steps once more, untimed, with the clock reader running (phase A is bound by vector issue: its time follows the
        # engine clock, and boxes of the pool differ by ~10 % on it)
        from tools.clocks import ClockSampler
        cs = ClockSampler(device)
        cs.start()
        for _ in range(n):
            e.process_batch_device(pool[t_seen % T], fg, None, None)
            t_seen += 1
        torch.cuda.synchronize()
        clk = cs.stop()
        r = _leg(wall_ms, px, 110)
        r["ms_per_step_wall"] = r.pop("kernel_avg_ms")
        r.update({"frames_1080p_per_s": round(S / (wall_ms * 1e-3), 1), "dominant_kernel": kname, "dominant_kernel_avg_ms": round(ms, 4),
                  "foreground_ratio": round(float((fg != 0).float().mean()), 4),
                  "clocks_during_the_same_steps_repeated": {k: clk.get(k) for k in ("engine_clock_MHz", "board_power_W", "samples", "error") if k in clk}})
        return r
    for _ in range(5):
        e.process_batch_device(pool[t_seen % T], fg, None, None)
        t_seen += 1
    c3["young_model"] = dict(timed(30), model_age_frames=6)
    while t_seen < 300:
        e.process_batch_device(src.into(cur), fg, None, None)
        t_seen += 1
    pool[:30] = src.pool(30)
    t_seen = T * 9  # (pool index 0 again: the 30 timed aged steps read 30 frames the model has never seen)
    c3["aged_model"] = dict(timed(30), model_age_frames=300)
    c3["aged_model"]["as_two_ranges_on_two_hip_streams"] = timed_two_ranges(30)
    # model initialisation again (every stream reset: the next frame runs SuBSENSE::initialize - LBSP of the frame, refreshModel(1.0):
    # 50 samples x 16 bytes per pixel written - and then the ordinary step), buffers already allocated
    for s_ in range(S):
        e.reset_stream(s_)
    torch.cuda.synchronize()
    w0 = time.perf_counter()
    e.process_batch_device(pool[0], fg, None, None)
    torch.cuda.synchronize()
    reinit_ms = (time.perf_counter() - w0) * 1e3
    model_bytes = S * rows * cols * 50 * 16
    c3["model_initialisation"] = {"first_frame_ms_buffers_allocated": round(reinit_ms, 3), "model_bytes_written": model_bytes,
                                  "note": "one call: initialise + the first ordinary step (~1.7 ms of it); the full refresh writes every record of the model once (ss_refresh_kernel; "
                                          "no separate clear since round 4) - compare calibration.fill_GBps_plain"}
    t_seen = 1
    for _ in range(5):
        e.process_batch_device(pool[t_seen % T], fg, None, None)
        t_seen += 1
    c3["young_model"]["as_two_ranges_on_two_hip_streams"] = timed_two_ranges(30)  # (the re-initialised model at the same age as the leg above)
    e.close()
    from oracle import pyoracle
    lut = pyoracle.lbsp_lut(0.333, 0, 3)
    from tracking_amd.engine import lbsp_describe_batch_device
    imgs = pool[0, :min(S, 16)].contiguous()
    for _ in range(3):
        lbsp_describe_batch_device(imgs, lut)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        lbsp_describe_batch_device(imgs, lut)
    b.record()
    torch.cuda.synchronize()
    c3["lbsp_kernel"] = dict(_leg(a.elapsed_time(b) / 20, imgs.shape[0] * rows * cols, 9), kernel="lbsp_kernel",
                             note="%d frames per launch (bgs_lbsp_describe_batch_device), torch events around 20 calls incl. the output allocation; r 3 + w 6 B/pixel" % imgs.shape[0])
    if cpu:
        sample = pool[:3, 0].cpu().numpy()
        rate, th = cpu_rate(capi.SUBSENSE, sample, warm=1)
        c3["cpu_baseline"] = {"value": round(rate, 3), "unit": "Mpixels/s", "cores": th, "kind": "port",
                              "sample": "2 timed 1920x1080 frames of stream 0 after the initialising frame, oracle SuBSENSE (oracle/subsense_oracle.c), 1 thread"}
    out["configs3_subsense_1080p"] = c3
    del pool
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=8)
    ap.add_argument("--only", default="", help="subsense: just the SuBSENSE lines")
    args = ap.parse_args()
    S = args.streams
    if args.only == "subsense":
        run_subsense(2)
        run_subsense(2, kind="smooth")
        return
    if args.only == "byte":  # the byte-stream kernels only, no CPU leg (kernel iteration)
        run(capi.WMV, "WeightedMovingVarianceBGS", 2160, 3840, S, 10, cpu_frames=0)
        run(capi.WMV, "WeightedMovingVarianceBGS (S_sat: every pixel moves)", 2160, 3840, S, 10, cpu_frames=0, kind="sat")
        run(capi.ABL, "AdaptiveBackgroundLearning", 2160, 3840, S, 10, borrow=False, cpu_frames=0)
        run(capi.ABL, "AdaptiveBackgroundLearning (S_sat)", 2160, 3840, S, 10, borrow=False, cpu_frames=0, kind="sat")
        run(capi.WMM, "WeightedMovingMeanBGS (+bg)", 2160, 3840, S, 13, want_bg=True, cpu_frames=0)
        run(capi.WMM, "WeightedMovingMeanBGS (mask only)", 2160, 3840, S, 10, cpu_frames=0)
        run(capi.FRAME_DIFF, "FrameDifferenceBGS", 2160, 3840, S, 7, cpu_frames=0)
        run(capi.SIGMA_DELTA, "SigmaDeltaBGS", 2160, 3840, S, 16, borrow=False, cpu_frames=0)
        run(capi.ASBL, "AdaptiveSelectiveBackgroundLearning", 2160, 3840, S, 6, borrow=False, cpu_frames=0)
        det = capi.default_params(capi.ASBL)
        det.learning_frames = 4
        run(capi.ASBL, "ASBL (detection phase)", 2160, 3840, S, 6, borrow=False, cpu_frames=0, params=det)
        return
    if args.only in ("clip8sat", "clip8surv"):  # one short leg, for counter passes
        run_clip(kind=args.only[5:], steps=6, Ts=(8,))
        return
    if args.only == "clip":
        run_clip(kind="sat")
        run_clip(kind="surv")
        return
    if args.only == "clipdp":  # package_bgs/dp GMMs (K = 3: 60 / 72 B/px of model + the count byte)
        run_clip(S=16, kind="sat", algo=capi.DP_ZIVKOVIC_AGMM, dense=122.0, label="DPZivkovicAGMM")
        run_clip(S=16, kind="surv", algo=capi.DP_ZIVKOVIC_AGMM, dense=122.0, label="DPZivkovicAGMM")
        run_clip(S=16, kind="sat", algo=capi.DP_GRIMSON_GMM, dense=146.0, label="DPGrimsonGMM")
        return
    if args.only == "clipfd":  # history classes: inside a clip the clip's own frames are the history (no per-frame copy into the ring)
        for algo, label in ((capi.FRAME_DIFF, "FrameDifference"), (capi.WMV, "WeightedMovingVariance")):
            run_clip(S=8, rows=2160, cols=3840, kind="surv", algo=algo, dense=3.0, label=label, Ts=(1, 4, 8))
        return
    if args.only == "clip1":  # MixtureOfGaussianV1BGS clips (16 streams: 320 B/px of model)
        run_clip(S=16, kind="sat", algo=capi.MOG1, dense=320.0, label="MOG1")
        run_clip(S=16, kind="surv", algo=capi.MOG1, dense=320.0, label="MOG1")
        return
    if args.only == "byte32":  # the same byte-stream kernels with 32 x 4K per launch (2.6 GB): how much of the gap to 0.79 is launch size
        run(capi.WMV, "WeightedMovingVarianceBGS", 2160, 3840, 32, 10, cpu_frames=0, steps=30)
        run(capi.ABL, "AdaptiveBackgroundLearning", 2160, 3840, 32, 10, borrow=False, cpu_frames=0, steps=30)
        run(capi.FRAME_DIFF, "FrameDifferenceBGS", 2160, 3840, 32, 7, cpu_frames=0, steps=30)
        return
    if args.only in ("mog1", "mog1sat", "mog1surv"):  # MixtureOfGaussianV1BGS update kernel on both inputs (no CPU leg: kernel iteration / counter passes)
        if args.only != "mog1surv":
            run(capi.MOG1, "MixtureOfGaussianV1BGS (S_sat)", 1080, 1920, 16, 324, borrow=False, kind="sat", cpu_frames=0)
        if args.only != "mog1sat":
            run(capi.MOG1, "MixtureOfGaussianV1BGS (S_surv)", 1080, 1920, 16, 324, borrow=False, cpu_frames=0)
        return
    if args.only == "lbsp":
        run_lbsp()
        return
    if args.only == "pipeline":
        run_pipeline()
        return
    if args.only == "group":  # BASELINE configs[2] as one fused launch (bgs_group of WMV + ABL, kernel_fanout.h) - for counter passes
        from tracking_amd.engine import Group
        dev = torch.device("cuda", 0)
        rows, cols, T = 2160, 3840, 8
        pool = torch.empty((T, S, rows, cols, 3), dtype=torch.uint8, device=dev)
        for s_ in range(S):
            pool[:, s_] = synth.s_surv(T, rows, cols, seed=4321 + s_, device=dev)
        fg1, fg2 = (torch.empty((S, rows, cols), dtype=torch.uint8, device=dev) for _ in range(2))
        grp = Group([capi.WMV, capi.ABL], device=0, n_streams=S)
        grp.set_geometry(rows, cols, 3)
        grp.set_option(capi.OPT_BORROW_FRAMES, 1)
        for t in range(10):
            grp.process_batch_device(pool[t % T], [fg1, fg2], None)
        torch.cuda.synchronize()
        grp.enable_kernel_timing(True)
        for t in range(40):
            grp.process_batch_device(pool[(10 + t) % T], [fg1, fg2], None)
        torch.cuda.synchronize()
        ms, _ = grp.kernel_timing()
        px = S * rows * cols
        print("WMV + ABL as one bgs_group       %dx%d x%d streams: kernel fan_kernel %.4f ms -> %.1f Mpix/s  %.1f GB/s algorithmic (17 B/px) = %.1f%% of 8 TB/s"
              % (cols, rows, S, ms, px / ms / 1e3, 17 * px / ms / 1e6, 17 * px / ms / 1e6 / 80.0))
        grp.close()
        return
    if args.only in ("gmg", "gmgsat"):  # GMG in normal operation (counter passes / kernel iteration)
        pg = capi.default_params(capi.GMG)
        pg.gmg_init_frames = 4
        run(capi.GMG, "GMG (data-dependent traffic)", 1080, 1920, 8, 16, borrow=False, cpu_frames=0, params=pg, kind="sat" if args.only == "gmgsat" else "surv")
        return
    if args.only == "dp":
        run_dp()
        return
    if args.only in ("dpzsat", "dpzsurv", "dpgsat", "dpgsurv"):  # one dp GMM leg (counter passes)
        algo, bpp, nm = (capi.DP_ZIVKOVIC_AGMM, 126, "DPZivkovicAGMMBGS") if args.only[2] == "z" else (capi.DP_GRIMSON_GMM, 150, "DPGrimsonGMMBGS")
        kind = args.only[3:]
        run(algo, "%s (K=3, S_%s)" % (nm, kind), 1080, 1920, 16, bpp, borrow=False, kind=kind, cpu_frames=0)
        return
    if args.only == "cc":
        run_cc()
        return
    if args.only == "lobster":
        run_subsense(8, algo=capi.LOBSTER, label="LOBSTERBGS")
        run_subsense(8, kind="smooth", algo=capi.LOBSTER, label="LOBSTERBGS")
        return
    if args.only == "subsense8":
        run_subsense(8)
        return
    if args.only == "driverconfigs":  # the configs[2] / configs[3] block exactly as bench.py puts it into the driver's line (no CPU legs): A/B scripts
        out = configs_block(S=S, cpu=False)
        c3 = out["configs3_subsense_1080p"]
        for k in ("young_model", "aged_model"):
            clk = c3[k].get("clocks_during_the_same_steps_repeated") or {}
            print("driver configs3 SuBSENSE %-11s %.4f ms/step wall, phase A %.4f ms, fg ratio %.4f; engine clock %s MHz, board power %s W" % (
                k, c3[k]["ms_per_step_wall"], c3[k]["dominant_kernel_avg_ms"], c3[k]["foreground_ratio"], (clk.get("engine_clock_MHz") or {}).get("median"), (clk.get("board_power_W") or {}).get("median")))
        return
    if args.only == "subsense8both":  # young and aged model in one process (A/B scripts)
        run_subsense(8)
        run_subsense(8, warm=300)
        return
    if args.only == "subsense8x2":  # the 8 streams as two / four ranges on their own HIP streams
        run_subsense(8)
        run_subsense(8, groups=2)
        run_subsense(8, groups=4)
        run_subsense(8, kind="smooth", groups=2)
        return
    if args.only == "subsense8agedx2":  # aged model: phase B (sample writes, HBM-bound) of one range beside phase A (VALU-bound) of the other
        run_subsense(8, warm=300)
        run_subsense(8, warm=300, groups=2)
        run_subsense(8, warm=300, groups=4)
        run_subsense(8, warm=300, groups=8)
        return
    if args.only == "subsense8aged1":  # one leg, for counter passes
        run_subsense(8, warm=300, steps=10)
        return
    if args.only == "subsense8aged":  # the model after 300 frames: update rates have settled, far fewer sample writes per frame
        run_subsense(8, warm=300)
        run_subsense(8, kind="smooth", warm=300)
        return
    run(capi.WMV, "WeightedMovingVarianceBGS", 2160, 3840, S, 10)
    run(capi.ABL, "AdaptiveBackgroundLearning", 2160, 3840, S, 10, borrow=False)
    run(capi.WMM, "WeightedMovingMeanBGS (+bg)", 2160, 3840, S, 13, want_bg=True)
    run(capi.FRAME_DIFF, "FrameDifferenceBGS", 2160, 3840, S, 7)
    run(capi.STATIC_FRAME_DIFF, "StaticFrameDifferenceBGS", 2160, 3840, S, 7, borrow=False)
    run(capi.SIGMA_DELTA, "SigmaDeltaBGS", 2160, 3840, S, 16, borrow=False)
    run(capi.ASBL, "AdaptiveSelectiveBackgroundLearning", 2160, 3840, S, 6, borrow=False)
    run(capi.MOG1, "MixtureOfGaussianV1BGS", 1080, 1920, 16, 324, borrow=False)
    pg = capi.default_params(capi.GMG)
    pg.gmg_init_frames = 4  # so the timed frames are normal-operation frames
    run(capi.GMG, "GMG (data-dependent traffic)", 1080, 1920, 8, 16, borrow=False, cpu_frames=8, cpu_warm=5, params=pg)
    run_subsense(2)
    run_subsense(2, kind="smooth")
    run_lbsp()

if __name__ == "__main__":
    main()
