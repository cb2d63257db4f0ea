"""Synthetic camera streams for tests and bench.py.  Data generation only.

S_sat  : every pixel cycles through 5 levels 40 apart (> sqrt(Tg*varMax) = 26) so MOG2 settles into 5 live modes per
         pixel -> the dense 206 B/pixel traffic model holds exactly (roofline workload).
S_surv : static random background + N(0,3^2) sensor noise + 6 opaque moving rectangles (surveillance-like workload).
Both are produced with torch ops so they can be generated directly in HBM (device='cuda') or on the host.
"""
import numpy as np
import torch


def s_sat(n_frames, rows, cols, seed=1234, device="cpu", t0=0):
    """uint8 [n_frames][rows][cols][3]"""
    g = torch.Generator(device=device).manual_seed(seed)
    base = torch.randint(20, 61, (rows, cols, 3), generator=g, device=device, dtype=torch.int16)
    phi = torch.randint(0, 5, (rows, cols, 1), generator=g, device=device, dtype=torch.int16)
    out = torch.empty((n_frames, rows, cols, 3), dtype=torch.uint8, device=device)
    for t in range(n_frames):
        noise = torch.randint(-2, 3, (rows, cols, 3), generator=g, device=device, dtype=torch.int16)
        out[t] = (base + 40 * ((t0 + t + phi) % 5) + noise).clamp_(0, 255).to(torch.uint8)
    return out


def s_surv(n_frames, rows, cols, seed=4321, device="cpu", t0=0, box=(120, 200), n_boxes=6, speed=4):
    """uint8 [n_frames][rows][cols][3]"""
    g = torch.Generator(device=device).manual_seed(seed)
    bg = torch.randint(0, 256, (rows, cols, 3), generator=g, device=device, dtype=torch.int16)
    cg = torch.Generator(device="cpu").manual_seed(seed + 1)
    bh, bw = min(box[0], max(1, rows // 3)), min(box[1], max(1, cols // 3))
    pos = torch.stack([torch.randint(0, max(1, rows - bh), (n_boxes,), generator=cg), torch.randint(0, max(1, cols - bw), (n_boxes,), generator=cg)], 1)
    vel = torch.randint(0, 2, (n_boxes, 2), generator=cg) * 2 - 1
    col = torch.randint(0, 256, (n_boxes, 3), generator=cg).to(torch.int16)
    out = torch.empty((n_frames, rows, cols, 3), dtype=torch.uint8, device=device)
    for t in range(n_frames):
        noise = (torch.randn((rows, cols, 3), generator=g, device=device) * 3.0).round().to(torch.int16)
        f = (bg + noise).clamp_(0, 255)
        for b in range(n_boxes):
            step = (t0 + t) * speed
            y = int((pos[b, 0] + vel[b, 0] * step) % max(1, rows - bh))
            x = int((pos[b, 1] + vel[b, 1] * step) % max(1, cols - bw))
            f[y:y + bh, x:x + bw] = col[b].to(device)
        out[t] = f.to(torch.uint8)
    return out


def s_smooth(n_frames, rows, cols, seed=777, device="cpu", t0=0, n_boxes=6, speed=4):
    """Like s_surv but with a smooth (low-frequency) static background instead of per-pixel random texture: closer to real video,
    where SuBSENSE's sample-consensus loop exits after 2-3 samples."""
    g = torch.Generator(device=device).manual_seed(seed)
    low = torch.rand((1, 3, max(2, rows // 32), max(2, cols // 32)), generator=g, device=device) * 200 + 20
    bg = torch.nn.functional.interpolate(low, size=(rows, cols), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).round().to(torch.int16)
    cg = torch.Generator(device="cpu").manual_seed(seed + 1)
    bh, bw = min(120, max(1, rows // 3)), min(200, max(1, cols // 3))
    pos = torch.stack([torch.randint(0, max(1, rows - bh), (n_boxes,), generator=cg), torch.randint(0, max(1, cols - bw), (n_boxes,), generator=cg)], 1)
    vel = torch.randint(0, 2, (n_boxes, 2), generator=cg) * 2 - 1
    col = torch.randint(0, 256, (n_boxes, 3), generator=cg).to(torch.int16)
    out = torch.empty((n_frames, rows, cols, 3), dtype=torch.uint8, device=device)
    for t in range(n_frames):
        noise = (torch.randn((rows, cols, 3), generator=g, device=device) * 2.0).round().to(torch.int16)
        f = (bg + noise).clamp_(0, 255)
        for b in range(n_boxes):
            step = (t0 + t) * speed
            y = int((pos[b, 0] + vel[b, 0] * step) % max(1, rows - bh))
            x = int((pos[b, 1] + vel[b, 1] * step) % max(1, cols - bw))
            f[y:y + bh, x:x + bw] = col[b].to(device)
        out[t] = f.to(torch.uint8)
    return out


def numpy_frames(kind, n_frames, rows, cols, seed):
    fn = {"sat": s_sat, "surv": s_surv, "smooth": s_smooth}[kind]
    return fn(n_frames, rows, cols, seed=seed).numpy()


def random_frames(n_frames, rows, cols, channels=3, seed=0, smooth=True):
    """Small seeded test clips: a slowly varying scene (so models actually match) plus sparse jumps."""
    rng = np.random.default_rng(seed)
    shape = (rows, cols, channels) if channels > 1 else (rows, cols)
    base = rng.integers(0, 256, shape).astype(np.int32)
    out = []
    for t in range(n_frames):
        f = base + rng.integers(-6, 7, shape)
        jump = rng.random(shape[:2]) < 0.15
        jv = rng.integers(0, 256, shape)
        f = np.where(jump[..., None] if channels > 1 else jump, jv, f) if smooth else rng.integers(0, 256, shape)
        out.append(np.clip(f, 0, 255).astype(np.uint8))
    return np.stack(out)


# ---- batched, stateful sources for bench.py: every stream of a GPU in one set of torch ops, FRESH noise on every call -------------
# (round 3's bench cycled a pool of 10 / 16 frames: every pixel then met the same few noise values per level for ever, and a model
# that has seen a value before rewrites less than one that meets fresh noise - the verdict's point.  A source draws new noise for
# every frame it produces; bench.py generates the untimed phases frame by frame and the timed phase into a pool of DISTINCT frames
# that is resident in HBM before the clock starts.)
class _Streams:
    def __init__(self, streams, rows, cols, seed0, device):
        self.S, self.rows, self.cols, self.device = streams, rows, cols, device
        self.g = torch.Generator(device=device).manual_seed(seed0)
        self.t = 0

    def into(self, out):
        """next frame of every stream -> out [S][rows][cols][3] uint8 (a torch tensor on self.device)"""
        out.copy_(self._frame(self.t))
        self.t += 1
        return out

    def pool(self, n):
        """n consecutive DISTINCT frames [n][S][rows][cols][3]"""
        out = torch.empty((n, self.S, self.rows, self.cols, 3), dtype=torch.uint8, device=self.device)
        for i in range(n):
            self.into(out[i])
        return out


class SatStreams(_Streams):
    """S_sat: per pixel and channel a base level U{20..60}, frame t shows base + 40 ((t + phi) mod 5) + U{-2..2}: five
    modes 40 grey levels apart, all live."""
    STEP, NOISE, BASE = 40, 2, (20, 61)

    def __init__(self, streams, rows, cols, seed0=1234, device="cpu"):
        super().__init__(streams, rows, cols, seed0, device)
        self.base = torch.randint(self.BASE[0], self.BASE[1], (streams, rows, cols, 3), generator=self.g, device=device, dtype=torch.int16)
        self.phi = torch.randint(0, 5, (streams, rows, cols, 1), generator=self.g, device=device, dtype=torch.int16)

    def _frame(self, t):
        noise = torch.randint(-self.NOISE, self.NOISE + 1, (self.S, self.rows, self.cols, 3), generator=self.g, device=self.device, dtype=torch.int16)
        return (self.base + self.STEP * ((t + self.phi) % 5) + noise).clamp_(0, 255).to(torch.uint8)


class DenseStreams(SatStreams):
    """S_dense, the WORST case of the summary filter: five modes only 8 grey levels apart with U{-1..1} noise.  8 levels in every channel
    is dist2 = 192 > Tg var = 36: five separate modes, all live - but far inside the ~13 levels per channel a 4-byte summary needs to
    prove a mode out (kernel_mog2.h mog2_reject), so every record of every pixel has to be read."""
    STEP, NOISE, BASE = 8, 1, (20, 201)


class SurvStreams(_Streams):
    """S_surv: static background U{0..255}, per-frame sensor noise N(0, 3^2), 6 opaque 120 x 200 rectangles of uniform
    colour moving 4 px per frame."""

    def __init__(self, streams, rows, cols, seed0=4321, device="cpu", box=(120, 200), n_boxes=6, speed=4):
        super().__init__(streams, rows, cols, seed0, device)
        self.bg = torch.randint(0, 256, (streams, rows, cols, 3), generator=self.g, device=device, dtype=torch.int16)
        cg = torch.Generator(device="cpu").manual_seed(seed0 + 1)
        self.bh, self.bw = min(box[0], max(1, rows // 3)), min(box[1], max(1, cols // 3))
        self.pos = torch.stack([torch.randint(0, max(1, rows - self.bh), (streams, n_boxes), generator=cg), torch.randint(0, max(1, cols - self.bw), (streams, n_boxes), generator=cg)], 2)
        self.vel = torch.randint(0, 2, (streams, n_boxes, 2), generator=cg) * 2 - 1
        self.col = torch.randint(0, 256, (streams, n_boxes, 3), generator=cg).to(torch.int16).to(device)
        self.speed, self.n_boxes = speed, n_boxes

    def _frame(self, t):
        noise = (torch.randn((self.S, self.rows, self.cols, 3), generator=self.g, device=self.device) * 3.0).round_().to(torch.int16)
        f = (self.bg + noise).clamp_(0, 255)
        step = t * self.speed
        for s in range(self.S):
            for b in range(self.n_boxes):
                y = int((self.pos[s, b, 0] + self.vel[s, b, 0] * step) % max(1, self.rows - self.bh))
                x = int((self.pos[s, b, 1] + self.vel[s, b, 1] * step) % max(1, self.cols - self.bw))
                f[s, y:y + self.bh, x:x + self.bw] = self.col[s, b]
        return f.to(torch.uint8)
