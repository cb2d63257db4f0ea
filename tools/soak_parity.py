"""Long-clip parity soak (GPU box): SuBSENSE / LOBSTER / MOG2-clips against the CPU oracle over a few hundred QVGA frames with scene
cuts and moving boxes - more frames than the test-suite affords.  Prints one line per leg; exits non-zero on the first difference."""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")
from gpu_helpers import check_mog2_state, check_subsense_state  # noqa: E402
from oracle import pyoracle  # noqa: E402
from tools import synth  # noqa: E402
from tracking_amd import Engine, capi  # noqa: E402


def clip(n, H, W):
    parts = [synth.numpy_frames("smooth", n // 3, H, W, seed=1), 255 - synth.numpy_frames("surv", n // 3, H, W, seed=2) // 3,
             synth.numpy_frames("surv", n - 2 * (n // 3), H, W, seed=3)]
    return np.concatenate(parts)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 240
    H, W = 240, 320
    frames = clip(n, H, W)
    for algo, name in ((capi.SUBSENSE, "SuBSENSE"), (capi.LOBSTER, "LOBSTER")):
        eng, orc = Engine(algo), pyoracle.Oracle(algo)
        fgsum = 0
        for t, f in enumerate(frames):
            fg, bg = eng.process(f)
            ofg, obg = orc.process(f)
            assert np.array_equal(fg, ofg), "%s frame %d: %d mask pixels differ" % (name, t, int((fg != ofg).sum()))
            assert np.array_equal(bg, obg), "%s frame %d: background differs" % (name, t)
            fgsum += int((fg != 0).sum())
        if algo == capi.SUBSENSE:
            check_subsense_state(eng, orc, H, W)
        print("%s: %d frames %dx%d bit-exact (masks, backgrounds%s); mean foreground %.2f %%" % (name, n, W, H, ", whole model" if algo == capi.SUBSENSE else "", 100.0 * fgsum / (n * H * W)), flush=True)
    import torch
    S = 2
    clips = np.stack([frames, frames[::-1].copy()])  # [S][T][H][W][3]
    eng = Engine(capi.MOG2, n_streams=S)
    eng.set_geometry(H, W, 3)
    orcs = [pyoracle.Oracle(capi.MOG2) for _ in range(S)]
    t0 = 0
    while t0 < n:
        k = min(13, n - t0)  # 8 + 4 + 1
        d = torch.from_numpy(np.ascontiguousarray(clips[:, t0:t0 + k].transpose(1, 0, 2, 3, 4))).cuda()
        fg = torch.zeros((k, S, H, W), dtype=torch.uint8, device="cuda")
        eng.process_clip_device(d, k, fg)
        torch.cuda.synchronize()
        fgh = fg.cpu().numpy()
        for j in range(k):
            for s in range(S):
                ofg, _ = orcs[s].process(clips[s, t0 + j], want_bg=False)
                assert np.array_equal(fgh[j, s], ofg), "MOG2 clip frame %d stream %d" % (t0 + j, s)
        t0 += k
    for s in range(S):
        check_mog2_state(eng, orcs[s], H * W, stream=s)
    print("MOG2 clips: %d frames x %d streams bit-exact (masks), model within 1e-4" % (n, S), flush=True)


if __name__ == "__main__":
    main()
