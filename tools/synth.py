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
