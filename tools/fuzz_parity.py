"""Randomised differential test of the HIP path against the CPU oracle (GPU box): random class, geometry (odd sizes included),
parameters inside the ranges the engine accepts, number of streams, and a random mix of entry points (host frames, device
batches, device ranges, clips).  Every mask, every background the class delivers and - where the helpers know the model - the
state at the end must match.  Usage: python tools/fuzz_parity.py [seconds] [seed]; prints one line per case, exits 1 on the
first mismatch with the seed that reproduces it.  Extra words: "big" = frames of several tiles (33x64 .. 200x320), "v" = print each case
before it runs."""
import sys
import time

import numpy as np

ROOT = __file__.rsplit("/", 2)[0]
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT + "/tests")
import torch  # noqa: E402
from gpu_helpers import ALGOS, check_state, check_subsense_state  # noqa: E402
from oracle import pyoracle  # noqa: E402
from tools import synth  # noqa: E402
from tracking_amd import Engine, capi  # noqa: E402

VERBOSE = False
BIG = False
ALL = dict(ALGOS)
ALL["SuBSENSEBGS"] = capi.SUBSENSE
ALL["LOBSTERBGS"] = capi.LOBSTER


def rand_params(rng, name, algo):
    p = capi.default_params(algo)
    kw = {}
    if rng.random() < 0.3:
        kw["threshold"] = int(rng.integers(0, 60))
    if rng.random() < 0.15 and name not in ("SuBSENSEBGS", "LOBSTERBGS", "GMG"):
        kw["enable_threshold"] = 0
    if name in ("MixtureOfGaussianV2BGS", "MixtureOfGaussianV1BGS", "AdaptiveBackgroundLearning"):
        kw["alpha"] = float(rng.choice([-1.0, 0.001, 0.01, 0.05, 0.2, 0.7])) if name != "AdaptiveBackgroundLearning" else float(rng.choice([0.01, 0.05, 0.3]))
    if name == "MixtureOfGaussianV2BGS":
        if rng.random() < 0.5:
            kw.update(mog2_var_threshold=float(rng.choice([4, 16, 36])), mog2_background_ratio=float(rng.choice([0.5, 0.9])), mog2_ct=float(rng.choice([0.0, 0.05, 0.3])),
                      mog2_history=int(rng.choice([3, 20, 500])), mog2_detect_shadows=int(rng.integers(0, 2)))
    if name == "MixtureOfGaussianV1BGS" and rng.random() < 0.5:
        kw.update(mog1_history=int(rng.choice([2, 10, 200])), mog1_background_ratio=float(rng.choice([0.4, 0.7, 0.95])), mog1_noise_sigma=float(rng.choice([5.0, 15.0])))
    if name in ("WeightedMovingMeanBGS", "WeightedMovingVarianceBGS"):
        kw["enable_weight"] = int(rng.integers(0, 2))
    if name == "AdaptiveSelectiveBackgroundLearning":
        kw.update(learning_frames=int(rng.choice([2, 5, 90])), alpha_learn=float(rng.choice([0.05, 0.3])), alpha_detection=float(rng.choice([0.05, 0.2])))
    if name == "SigmaDeltaBGS":
        kw.update(sd_amp_factor=int(rng.choice([1, 2, 4, 300])), sd_min_var=int(rng.choice([1, 15])), sd_max_var=int(rng.choice([100, 255])))
    if name.startswith("DP"):
        kw.update(dp_threshold=float(rng.choice([9.0, 16.0, 40.0])), dp_alpha=float(rng.choice([1e-4, 0.01, 0.3])), dp_gaussians=int(rng.integers(1, 6)),
                  dp_sampling_rate=int(rng.choice([2, 7])), learning_frames=int(rng.choice([3, 30])))
    if name in ("SuBSENSEBGS", "LOBSTERBGS"):
        ns = int(rng.choice([3, 7, 20, 35, 50]))
        kw.update(subsense_n_samples=ns, subsense_n_required=int(rng.integers(1, min(ns, 3) + 1)), lbsp_rel_threshold=float(rng.choice([0.2, 0.333, 0.365])),
                  subsense_min_color_dist_threshold=int(rng.choice([15, 30])), subsense_desc_dist_threshold_offset=int(rng.choice([1, 3, 4])))
    if name == "GMG":
        kw.update(gmg_init_frames=int(rng.choice([3, 8])), gmg_max_features=int(rng.choice([8, 64])), gmg_smoothing_radius=int(rng.choice([3, 7])))
    for k, v in kw.items():
        setattr(p, k, v)
    return p, kw


def frames_for(rng, T, H, W, seed):
    kind = rng.choice(["random", "surv", "smooth", "sat"])
    if kind == "random":
        f = synth.random_frames(T, H, W, 3, seed=seed)
    else:
        f = synth.numpy_frames(kind, T, H, W, seed=seed)
    if rng.random() < 0.3:  # a scene cut
        c = T // 2
        f = np.concatenate([f[:c], 255 - f[c:]])
    return np.ascontiguousarray(f)


def one_case(rng, case_seed):
    name = rng.choice(sorted(ALL))
    algo = ALL[name]
    small = name in ("SuBSENSEBGS", "LOBSTERBGS", "GMG")
    if BIG:  # several tiles in both directions, ragged edges
        H, W = int(rng.choice([33, 65, 97, 130, 200])), int(rng.choice([64, 97, 131, 257, 320]))
    else:
        H = int(rng.choice([5, 9, 16, 33, 48] if small else [1, 7, 16, 33, 64]))
        W = int(rng.choice([5, 37, 64, 131] if small else [3, 64, 70, 128, 200]))
    device_ok = True
    S = int(rng.integers(1, 4))
    T = int(rng.integers(4, 22))
    p, kw = rand_params(rng, name, algo)
    clips = np.stack([frames_for(rng, T, H, W, case_seed * 7 + s) for s in range(S)])  # [S][T][H][W][3]
    if VERBOSE:
        print("case %d: %s %dx%d x%d streams, %d frames, %s" % (case_seed, name, W, H, S, T, kw), flush=True)
    want_bits = (H * W) % 64 == 0
    try:
        eng = Engine(algo, params=p, n_streams=S)
        orcs = [pyoracle.Oracle(algo, params=p) for _ in range(S)]
    except capi.BgsError as e:
        return "%s rejected %s: %s" % (name, kw, e)
    mode = rng.choice(["host", "batch", "clip", "mixed"]) if device_ok else "host"
    try:
        eng.set_geometry(H, W, 3)
    except capi.BgsError as e:
        eng.close()
        return "%s %dx%d not supported: %s" % (name, W, H, str(e)[:60])
    if name == "MixtureOfGaussianV2BGS":  # round 3: which MOG2 kernel loads the model (dense / eager / count / auto / filter)
        level = int(rng.choice([0, 1, 2, 3, 4]))
        eng.set_option(capi.OPT_MOG2_SPARSE, level)
        kw["sparse"] = level
    resets = rng.random() < 0.35  # round 3: cameras of a batch are reset independently, so the streams of one call have different ages
    t = 0
    while t < T:
        if resets and t > 0 and rng.random() < 0.3:
            r = int(rng.integers(0, S))
            eng.reset_stream(r)
            orcs[r].close()
            orcs[r] = pyoracle.Oracle(algo, params=p)
        m = mode if mode != "mixed" else rng.choice(["host", "batch", "clip"])
        n = 1
        if m == "host":
            got = []
            for s in range(S):
                fg, bg = eng.process(clips[s, t], stream=s)
                got.append((fg, bg))
            outs = [[g] for g in got]
        else:
            n = 1 if m == "batch" else int(min(T - t, rng.integers(1, 12)))
            d = torch.from_numpy(np.ascontiguousarray(clips[:, t:t + n].transpose(1, 0, 2, 3, 4))).cuda()
            fg = torch.full((n, S, H, W), 7, dtype=torch.uint8, device="cuda")
            bits = torch.zeros((n, S, H * W // 64), dtype=torch.int64, device="cuda") if want_bits else None
            if m == "batch":
                flags = [eng.process_batch_device(d[0], fg[0], None, bits[0] if want_bits else None)]
            else:
                flags = eng.process_clip_device(d, n, fg, None, bits)
            torch.cuda.synchronize()
            fgh = fg.cpu().numpy()
            outs = [[(fgh[j, s], None) for j in range(n)] for s in range(S)]  # validity per stream: an untouched output keeps the marker 7
            if want_bits:
                bh = np.unpackbits(bits.cpu().numpy().view(np.uint8).reshape(n, S, -1), axis=2, bitorder="little").reshape(n, S, H, W)
        for s in range(S):
            for j in range(n):
                ofg, obg = orcs[s].process(clips[s, t + j])
                fg_j, bg_j = outs[s][j]
                if m == "host":
                    assert (fg_j is None) == (ofg is None), "%s frame %d stream %d: mask validity" % (name, t + j, s)
                elif ofg is None:
                    assert (fg_j == 7).all(), "%s frame %d stream %d (%s): a warm-up frame must leave the mask untouched" % (name, t + j, s, m)
                if ofg is not None:
                    assert np.array_equal(fg_j, ofg), "%s frame %d stream %d (%s): %d mask pixels differ" % (name, t + j, s, m, int((fg_j != ofg).sum()))
                    if m != "host" and want_bits:
                        assert np.array_equal(bh[j, s] * 255, np.where(ofg != 0, 255, 0)), "%s frame %d stream %d: packed mask" % (name, t + j, s)
                if m == "host":
                    assert (bg_j is None) == (obg is None), "%s frame %d: background validity" % (name, t + j)
                    if obg is not None:
                        assert np.array_equal(bg_j.reshape(obg.shape), obg), "%s frame %d stream %d: background differs" % (name, t + j, s)
        t += n
    for s in range(S):
        if name == "SuBSENSEBGS":
            check_subsense_state(eng, orcs[s], H, W, nS=p.subsense_n_samples, stream=s)
        elif name.startswith("DP"):
            from gpu_helpers import check_dp_state
            check_dp_state(name, eng, orcs[s], H * W, K=p.dp_gaussians, stream=s)
        elif name == "GMG":
            F, n = p.gmg_max_features, H * W
            assert np.array_equal(eng.get_state("nfeatures", (n,), np.int32, stream=s), orcs[s].get_state("nfeatures", (n,), np.int32)), "GMG feature counts"
            assert np.array_equal(eng.get_state("colors", (F, n), np.int32, stream=s), orcs[s].get_state("colors", (F, n), np.int32)), "GMG colours"
            assert float(np.max(np.abs(eng.get_state("weights", (F, n), np.float32, stream=s) - orcs[s].get_state("weights", (F, n), np.float32)))) <= 1e-4, "GMG weights"
        else:
            check_state(name, eng, orcs[s], H * W, stream=s)
    eng.close()
    return "%s %dx%d x%d streams, %d frames, %s%s %s: ok" % (name, W, H, S, T, mode, " +resets" if resets else "", kw)


def main():
    global VERBOSE, BIG
    VERBOSE = "v" in sys.argv[3:]
    BIG = "big" in sys.argv[3:]
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time()) % 100000
    t0 = time.time()
    k = 0
    while time.time() - t0 < budget:
        case_seed = seed0 + k
        rng = np.random.default_rng(case_seed)
        try:
            msg = one_case(rng, case_seed)
        except AssertionError as e:
            print("MISMATCH (seed %d): %s" % (case_seed, e), flush=True)
            sys.exit(1)
        print("[%d] %s" % (case_seed, msg), flush=True)
        k += 1
    print("%d cases in %.0f s, no mismatch" % (k, time.time() - t0))


if __name__ == "__main__":
    main()
