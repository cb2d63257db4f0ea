"""Engine — thin Python handle on a bgs_engine (include/bgs_hip.h), used by tests and bench.py.

Two call shapes, mirroring the two C entry points:
  process(frame)            host numpy frame in, (mask, background) out   == IBGS::process (package_bgs/IBGS.h:24)
  process_batch_device(...) torch CUDA tensors, asynchronous, all streams == the roofline path
numpy/torch only carry memory here; all arithmetic happens inside libbgs_hip.
"""
import ctypes as C

import numpy as np

from . import capi


class Engine:
    def __init__(self, algo, params=None, device=0, n_streams=1):
        self._h = C.c_void_p()
        self.algo = algo
        self.n_streams = n_streams
        self.params = params if params is not None else capi.default_params(algo)
        capi.check(capi.lib().bgs_create(algo, C.byref(self.params), device, n_streams, C.byref(self._h)))

    @classmethod
    def from_handle(cls, handle, algo, n_streams):
        """A view of a bgs_engine owned by someone else (a bgs_node's per-device engine): close() does not destroy it."""
        self = cls.__new__(cls)
        self._h, self.algo, self.n_streams, self.params, self._borrowed = handle, algo, n_streams, capi.default_params(algo), True
        return self

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            if not getattr(self, "_borrowed", False):
                capi.lib().bgs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- configuration ------------------------------------------------------
    def set_params(self, params):
        capi.check(capi.lib().bgs_set_params(self._h, C.byref(params)))
        self.params = params

    def set_option(self, option, value):
        capi.check(capi.lib().bgs_set_option(self._h, option, int(value)))

    def set_ingest(self, cfg):
        """bgs_set_ingest: process() then takes raw captured frames and returns outputs in the prepared geometry."""
        capi.check(capi.lib().bgs_set_ingest(self._h, C.byref(cfg) if cfg is not None else None))
        self._ingest = cfg

    def set_geometry(self, rows, cols, channels):
        capi.check(capi.lib().bgs_set_geometry(self._h, rows, cols, channels))

    # -- host path (IBGS::process) -------------------------------------------
    def process(self, frame, stream=0, want_bg=True):
        """frame: HxW or HxWxC uint8 (any row stride).  Returns (mask or None, background or None):
        None where the reference leaves the output untouched (warm-up frames, classes that write no background)."""
        if frame is None or frame.size == 0:
            flags = C.c_uint32(0)
            capi.check(capi.lib().bgs_process(self._h, stream, None, 0, 0, 3, 0, None, 0, None, 0, C.byref(flags)))
            return None, None
        assert frame.dtype == np.uint8
        rows, cols = frame.shape[:2]
        ch = 1 if frame.ndim == 2 else frame.shape[2]
        if frame.strides[-1] != 1 or (frame.ndim == 3 and frame.strides[1] != ch):
            frame = np.ascontiguousarray(frame)
        step = frame.strides[0]
        in_rows, in_cols = rows, cols
        if getattr(self, "_ingest", None) is not None:  # outputs come out in the prepared geometry
            r, c = C.c_int(0), C.c_int(0)
            capi.check(capi.lib().bgs_ingest_size(C.byref(self._ingest), rows, cols, C.byref(r), C.byref(c)))
            rows, cols = r.value, c.value
        fg = np.empty((rows, cols), np.uint8)
        bg_ch = 1 if self.algo == capi.ASBL else ch
        bg = np.empty((rows, cols, bg_ch), np.uint8) if want_bg else None
        flags = C.c_uint32(0)
        capi.check(capi.lib().bgs_process(
            self._h, stream, frame.ctypes.data_as(C.c_void_p), in_rows, in_cols, ch, step,
            fg.ctypes.data_as(C.c_void_p), cols,
            bg.ctypes.data_as(C.c_void_p) if bg is not None else None, cols * bg_ch, C.byref(flags)))
        f = flags.value
        if bg is not None and bg_ch == 1:
            bg = bg[:, :, 0]
        return (fg if f & capi.FG_VALID else None), (bg if (bg is not None and f & capi.BG_VALID) else None)

    def process_into(self, frame, fg, bg=None, stream=0):
        """bgs_process into caller-owned arrays (contiguous): what a caller that keeps its images allocated does; with
        OPT_HOST_REGISTER the same buffers are page-locked once and DMA'd in place.  Returns the out_flags."""
        rows, cols = frame.shape[:2]
        ch = 1 if frame.ndim == 2 else frame.shape[2]
        flags = C.c_uint32(0)
        capi.check(capi.lib().bgs_process(self._h, stream, frame.ctypes.data_as(C.c_void_p), rows, cols, ch, frame.strides[0], fg.ctypes.data_as(C.c_void_p), fg.strides[0],
                                          bg.ctypes.data_as(C.c_void_p) if bg is not None else None, bg.strides[0] if bg is not None else 0, C.byref(flags)))
        return flags.value

    def submit(self, frame, fg, bg=None, stream=0):
        """bgs_submit: queue one frame of `stream` (caller-owned contiguous arrays, valid until wait())."""
        rows, cols = frame.shape[:2]
        ch = 1 if frame.ndim == 2 else frame.shape[2]
        capi.check(capi.lib().bgs_submit(self._h, stream, frame.ctypes.data_as(C.c_void_p), rows, cols, ch, frame.strides[0], fg.ctypes.data_as(C.c_void_p), fg.strides[0],
                                         bg.ctypes.data_as(C.c_void_p) if bg is not None else None, bg.strides[0] if bg is not None else 0))

    def host_arena(self, array, on=True):
        """bgs_host_arena: page-lock a numpy array that holds the images of several cameras, once."""
        capi.check(capi.lib().bgs_host_arena(self._h, array.ctypes.data_as(C.c_void_p), array.nbytes, 1 if on else 0))

    def wait(self, stream=0):
        flags = C.c_uint32(0)
        capi.check(capi.lib().bgs_wait(self._h, stream, C.byref(flags)))
        return flags.value

    def process_mask_only_on_device(self, frame, stream=0):
        """bgs_process with fg = bg = NULL: the mask stays on the device (for last_mask_blobs).  Returns the out_flags."""
        rows, cols = frame.shape[:2]
        ch = 1 if frame.ndim == 2 else frame.shape[2]
        frame = np.ascontiguousarray(frame)
        flags = C.c_uint32(0)
        capi.check(capi.lib().bgs_process(self._h, stream, frame.ctypes.data_as(C.c_void_p), rows, cols, ch, frame.strides[0], None, 0, None, 0, C.byref(flags)))
        return flags.value

    def last_mask_blobs(self, stream=0, connectivity=8, min_w=0, min_h=0, max_boxes=4096):
        """bgs_last_mask_blobs: (boxes int32 [k][6] = BOX_FIELDS, moments int64 [k][4] = sx, sy, sxx, syy, count) of the mask the
        last process() call of `stream` left on the device; k = min(count, max_boxes)."""
        boxes = np.zeros((max(max_boxes, 1), 6), np.int32)
        mom = np.zeros((max(max_boxes, 1), 4), np.int64)
        n = C.c_int32(0)
        capi.check(capi.lib().bgs_last_mask_blobs(self._h, stream, connectivity, min_w, min_h, boxes.ctypes.data_as(C.c_void_p), mom.ctypes.data_as(C.c_void_p), max_boxes, C.byref(n)))
        k = min(n.value, max_boxes)
        return boxes[:k], mom[:k], n.value

    # -- device path -----------------------------------------------------------
    @staticmethod
    def _ptr(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    def process_batch_device(self, frames, fg=None, bg=None, fg_bits=None, hip_stream=None, first=None, count=None):
        """frames/fg/bg/fg_bits: torch CUDA tensors laid out as bgs_process_batch_device documents.
        Asynchronous on hip_stream (default: torch's current stream).  Returns the out_flags."""
        import torch
        if hip_stream is None:
            hip_stream = torch.cuda.current_stream().cuda_stream
        flags = C.c_uint32(0)
        l = capi.lib()
        if first is None:
            capi.check(l.bgs_process_batch_device(self._h, self._ptr(frames), self._ptr(fg), self._ptr(bg), self._ptr(fg_bits), C.c_void_p(hip_stream), C.byref(flags)))
        else:
            capi.check(l.bgs_process_range_device(self._h, first, count, self._ptr(frames), self._ptr(fg), self._ptr(bg), self._ptr(fg_bits), C.c_void_p(hip_stream), C.byref(flags)))
        return flags.value

    def process_clip_device(self, frames, nframes, fg=None, bg=None, fg_bits=None, hip_stream=None, first=0, count=None):
        """bgs_process_clip_device: frames [nframes][count][rows][cols][ch] (torch CUDA tensors, like the outputs).  Returns the
        nframes out_flags words."""
        import torch
        if hip_stream is None:
            hip_stream = torch.cuda.current_stream().cuda_stream
        flags = (C.c_uint32 * nframes)()
        if count is None:
            count = self.n_streams - first
        capi.check(capi.lib().bgs_process_clip_device(self._h, first, count, nframes, self._ptr(frames), self._ptr(fg), self._ptr(bg), self._ptr(fg_bits),
                                                      C.c_void_p(hip_stream), flags))
        return list(flags)

    # -- introspection -----------------------------------------------------------
    def get_state(self, plane, shape, dtype, stream=0):
        out = np.empty(shape, dtype)
        n = capi.lib().bgs_get_state(self._h, stream, plane.encode(), out.ctypes.data_as(C.c_void_p), out.nbytes)
        capi.check(n)
        assert n == out.nbytes, (n, out.nbytes)
        return out

    def frames_seen(self, stream=0):
        return capi.lib().bgs_frames_seen(self._h, stream)

    def reset_stream(self, stream):
        """bgs_reset_stream: the stream's next frame is a first frame again (model re-initialised on that call's HIP stream)."""
        capi.check(capi.lib().bgs_reset_stream(self._h, stream))

    def stream_flags(self, stream):
        f = C.c_uint32(0)
        capi.check(capi.lib().bgs_stream_flags(self._h, stream, C.byref(f)))
        return f.value

    def enable_kernel_timing(self, on=True):
        capi.check(capi.lib().bgs_enable_kernel_timing(self._h, 1 if on else 0))

    def kernel_timing(self):
        ms, n, name = C.c_double(0), C.c_int64(0), C.c_char_p()
        capi.check(capi.lib().bgs_kernel_timing(self._h, C.byref(ms), C.byref(n), C.byref(name)))
        return ms.value, n.value, (name.value or b"").decode()

    def kernel_timing_series(self, cap=16384):
        """Per-launch durations (ms) of the dominant kernel since timing was enabled."""
        buf = (C.c_float * cap)()
        n = capi.lib().bgs_kernel_timing_series(self._h, buf, cap)
        if n < 0:
            capi.check(int(n))
        return np.frombuffer(buf, dtype=np.float32, count=int(n)).copy()


# -- stand-alone device primitives ---------------------------------------------------------------------------------
def lbsp_describe_device(img, lut, out=None, device=0, hip_stream=None):
    """img: torch CUDA uint8 [rows][cols][C] (or [rows][cols]); lut: 256 uint8 thresholds (numpy).  Returns uint16 descriptors."""
    import torch
    rows, cols = img.shape[:2]
    ch = 1 if img.dim() == 2 else img.shape[2]
    if out is None:
        out = torch.empty((rows, cols, ch), dtype=torch.int16, device=img.device)
    if hip_stream is None:
        hip_stream = torch.cuda.current_stream().cuda_stream
    lut = np.ascontiguousarray(lut, dtype=np.uint8)
    capi.check(capi.lib().bgs_lbsp_describe_device(device, C.c_void_p(img.data_ptr()), rows, cols, ch, lut.ctypes.data_as(C.c_void_p),
                                                   C.c_void_p(out.data_ptr()), C.c_void_p(hip_stream)))
    return out


def lbsp_describe_batch_device(imgs, lut, device=0, hip_stream=None):
    """imgs: torch CUDA uint8 [images][rows][cols][C]; one launch for the whole stack.  Returns int16-typed uint16 descriptors."""
    import torch
    images, rows, cols, ch = imgs.shape
    out = torch.empty((images, rows, cols, ch), dtype=torch.int16, device=imgs.device)
    if hip_stream is None:
        hip_stream = torch.cuda.current_stream().cuda_stream
    lut = np.ascontiguousarray(lut, dtype=np.uint8)
    capi.check(capi.lib().bgs_lbsp_describe_batch_device(device, C.c_void_p(imgs.data_ptr()), images, rows, cols, ch, lut.ctypes.data_as(C.c_void_p),
                                                         C.c_void_p(out.data_ptr()), C.c_void_p(hip_stream)))
    return out


MORPH_ERODE, MORPH_DILATE, MORPH_MEDIAN, MORPH_MEDIAN_BINARY, MORPH_FLOODFILL_ORIGIN = 0, 1, 2, 3, 4


def mask_morph_device(src, op, ksize=3, iterations=1, device=0, hip_stream=None):
    """src: torch CUDA uint8 [rows][cols].  op: MORPH_ERODE / MORPH_DILATE (3x3, `iterations` times) or MORPH_MEDIAN (ksize)."""
    import torch
    rows, cols = src.shape
    dst = torch.empty_like(src)
    if hip_stream is None:
        hip_stream = torch.cuda.current_stream().cuda_stream
    capi.check(capi.lib().bgs_mask_morph_device(device, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), rows, cols, op, ksize, iterations, C.c_void_p(hip_stream)))
    return dst


BOX_FIELDS = ("x", "y", "w", "h", "area", "root")


def mask_components_device(mask, connectivity=8, max_boxes=4096, want_labels=True, device=0, hip_stream=None):
    """Connected components of a torch CUDA uint8 mask [rows][cols] (bgs_mask_components_device).
    Returns (labels int32 [rows][cols] or None, boxes int32 [min(count, max_boxes)][6] = BOX_FIELDS, count)."""
    import torch
    rows, cols = mask.shape
    if hip_stream is None:
        hip_stream = torch.cuda.current_stream().cuda_stream
    labels = torch.empty((rows, cols), dtype=torch.int32, device=mask.device) if want_labels else None
    boxes = torch.zeros((max(max_boxes, 1), 6), dtype=torch.int32, device=mask.device)
    count = torch.zeros(1, dtype=torch.int32, device=mask.device)
    work = torch.empty(capi.lib().bgs_mask_components_workspace(rows, cols), dtype=torch.uint8, device=mask.device)
    capi.check(capi.lib().bgs_mask_components_device(device, C.c_void_p(mask.data_ptr()), rows, cols, connectivity,
                                                     C.c_void_p(labels.data_ptr() if want_labels else 0), C.c_void_p(boxes.data_ptr()), max_boxes,
                                                     C.c_void_p(count.data_ptr()), C.c_void_p(work.data_ptr()), C.c_void_p(hip_stream)))
    n = int(count.item())
    return labels, boxes[:min(n, max_boxes)], n


def mask_components_batch_device(masks, connectivity=8, max_boxes=65536, want_labels=False, device=0, hip_stream=None):
    """Connected components of a stack of masks, torch CUDA uint8 [images][rows][cols], in one set of launches
    (bgs_mask_components_batch_device).  Returns (labels or None, boxes int32 [min(total, max_boxes)][6], offsets int32 [images+1]):
    image k owns boxes[offsets[k]:offsets[k+1]]."""
    import torch
    images, rows, cols = masks.shape
    if hip_stream is None:
        hip_stream = torch.cuda.current_stream().cuda_stream
    labels = torch.empty((images, rows, cols), dtype=torch.int32, device=masks.device) if want_labels else None
    boxes = torch.zeros((max(max_boxes, 1), 6), dtype=torch.int32, device=masks.device)
    offsets = torch.zeros(images + 1, dtype=torch.int32, device=masks.device)
    work = torch.empty(capi.lib().bgs_mask_components_batch_workspace(images, rows, cols), dtype=torch.uint8, device=masks.device)
    capi.check(capi.lib().bgs_mask_components_batch_device(device, C.c_void_p(masks.data_ptr()), images, rows, cols, connectivity,
                                                           C.c_void_p(labels.data_ptr() if want_labels else 0), C.c_void_p(boxes.data_ptr()), max_boxes,
                                                           C.c_void_p(offsets.data_ptr()), C.c_void_p(work.data_ptr()), C.c_void_p(hip_stream)))
    off = offsets.cpu()
    return labels, boxes[:min(int(off[-1]), max_boxes)], off


def mask_blobs_batch_device(masks, connectivity=8, max_boxes=65536, device=0, hip_stream=None):
    """bgs_mask_blobs_batch_device: boxes int32 [k][6], moments int64 [k][4] (sx, sy, sxx, syy), offsets int32 [images+1]."""
    import torch
    images, rows, cols = masks.shape
    if hip_stream is None:
        hip_stream = torch.cuda.current_stream().cuda_stream
    boxes = torch.zeros((max(max_boxes, 1), 6), dtype=torch.int32, device=masks.device)
    mom = torch.zeros((max(max_boxes, 1), 4), dtype=torch.int64, device=masks.device)
    offsets = torch.zeros(images + 1, dtype=torch.int32, device=masks.device)
    work = torch.empty(capi.lib().bgs_mask_components_batch_workspace(images, rows, cols), dtype=torch.uint8, device=masks.device)
    capi.check(capi.lib().bgs_mask_blobs_batch_device(device, C.c_void_p(masks.data_ptr()), images, rows, cols, connectivity, C.c_void_p(boxes.data_ptr()),
                                                      C.c_void_p(mom.data_ptr()), max_boxes, C.c_void_p(offsets.data_ptr()), C.c_void_p(work.data_ptr()), C.c_void_p(hip_stream)))
    off = offsets.cpu()
    k = min(int(off[-1]), max_boxes)
    return boxes[:k], mom[:k], off


# -- N3 frame preparation ------------------------------------------------------------------------------------------------
def ingest_device(cfg, frames, device=0, hip_stream=None):
    """bgs_ingest_device: frames torch CUDA uint8 [images][rows][cols][C] (or [images][rows][cols]); returns the prepared frames."""
    import torch
    images, rows, cols = frames.shape[:3]
    ch = 1 if frames.dim() == 3 else frames.shape[3]
    r, c = C.c_int(0), C.c_int(0)
    capi.check(capi.lib().bgs_ingest_size(C.byref(cfg), rows, cols, C.byref(r), C.byref(c)))
    out = torch.empty((images, r.value, c.value) if frames.dim() == 3 else (images, r.value, c.value, ch), dtype=torch.uint8, device=frames.device)
    ws = capi.lib().bgs_ingest_workspace(C.byref(cfg), images, rows, cols, ch)
    work = torch.empty(max(ws, 1), dtype=torch.uint8, device=frames.device)
    if hip_stream is None:
        hip_stream = torch.cuda.current_stream().cuda_stream
    capi.check(capi.lib().bgs_ingest_device(device, C.byref(cfg), C.c_void_p(frames.data_ptr()), images, rows, cols, ch, frames.stride(1),
                                            C.c_void_p(out.data_ptr()), C.c_void_p(work.data_ptr()) if ws else None, C.c_void_p(hip_stream)))
    return out


def ingest_host(cfg, frame, device=0):
    """bgs_ingest_host: frame numpy uint8 HxW or HxWxC (any row stride); returns the prepared frame."""
    rows, cols = frame.shape[:2]
    ch = 1 if frame.ndim == 2 else frame.shape[2]
    if frame.strides[-1] != 1 or (frame.ndim == 3 and frame.strides[1] != ch):
        frame = np.ascontiguousarray(frame)
    r, c = C.c_int(0), C.c_int(0)
    capi.check(capi.lib().bgs_ingest_size(C.byref(cfg), rows, cols, C.byref(r), C.byref(c)))
    out = np.empty((r.value, c.value) if frame.ndim == 2 else (r.value, c.value, ch), np.uint8)
    capi.check(capi.lib().bgs_ingest_host(device, C.byref(cfg), frame.ctypes.data_as(C.c_void_p), rows, cols, ch, frame.strides[0],
                                          out.ctypes.data_as(C.c_void_p), out.strides[0]))
    return out


class Group:
    """bgs_group: several classes on the same frames (include/bgs_hip.h, GROUPS) - the byte-stream classes among them as one fused
    kernel, the rest as member engines.  Binding only: no arithmetic here."""

    def __init__(self, algos, params=None, device=0, n_streams=1):
        self.algos = list(algos)
        self.n_streams = n_streams
        n = len(self.algos)
        self._params = [(p if p is not None else capi.default_params(a)) for a, p in zip(self.algos, params or [None] * n)]
        arr = (C.c_int * n)(*self.algos)
        pp = (C.c_void_p * n)(*[C.cast(C.byref(p), C.c_void_p) for p in self._params])
        self._h = C.c_void_p()
        capi.check(capi.lib().bgs_group_create(arr, pp, n, device, n_streams, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            capi.lib().bgs_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def is_fused(self, index):
        return capi.lib().bgs_group_is_fused(self._h, index) == 1

    def set_params(self, index, params):
        capi.check(capi.lib().bgs_group_set_params(self._h, index, C.byref(params)))
        self._params[index] = params

    def set_option(self, option, value):
        capi.check(capi.lib().bgs_group_set_option(self._h, option, int(value)))

    def set_geometry(self, rows, cols, channels):
        capi.check(capi.lib().bgs_group_set_geometry(self._h, rows, cols, channels))

    def process(self, frame, want_bg=True):
        """bgs_group_process (single-stream groups): returns [(mask or None, background or None)] per class, None where the
        reference leaves the output untouched."""
        assert frame.dtype == np.uint8
        rows, cols = frame.shape[:2]
        ch = 1 if frame.ndim == 2 else frame.shape[2]
        frame = np.ascontiguousarray(frame)
        n = len(self.algos)
        fgs = [np.empty((rows, cols), np.uint8) for _ in range(n)]
        bgs_ = [np.empty((rows, cols, 1 if a == capi.ASBL else ch), np.uint8) if want_bg else None for a in self.algos]
        pf = (C.c_void_p * n)(*[f.ctypes.data for f in fgs])
        pb = (C.c_void_p * n)(*[(b.ctypes.data if b is not None else None) for b in bgs_])
        flags = (C.c_uint32 * n)()
        capi.check(capi.lib().bgs_group_process(self._h, frame.ctypes.data_as(C.c_void_p), rows, cols, ch, frame.strides[0], pf, None, pb, None, flags))
        out = []
        for i in range(n):
            f = flags[i]
            b = bgs_[i]
            if b is not None and b.shape[2] == 1:
                b = b[:, :, 0]
            out.append((fgs[i] if f & capi.FG_VALID else None, b if (b is not None and f & capi.BG_VALID) else None))
        return out

    def process_batch_device(self, frames, fgs=None, bgs_=None, hip_stream=None):
        """bgs_group_process_batch_device: frames [n_streams][rows][cols][ch] (torch CUDA tensor), fgs / bgs_: lists of tensors or None
        per class.  Asynchronous on hip_stream (default: torch's current stream).  Returns the per-class out_flags."""
        import torch
        if hip_stream is None:
            hip_stream = torch.cuda.current_stream().cuda_stream
        n = len(self.algos)
        pf = (C.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for t in (fgs or [None] * n)])
        pb = (C.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for t in (bgs_ or [None] * n)])
        flags = (C.c_uint32 * n)()
        capi.check(capi.lib().bgs_group_process_batch_device(self._h, C.c_void_p(frames.data_ptr()), pf, pb, C.c_void_p(hip_stream), flags))
        return list(flags)

    def get_state(self, index, plane, shape, dtype, stream=0):
        out = np.empty(shape, dtype)
        n = capi.lib().bgs_group_get_state(self._h, index, stream, plane.encode(), out.ctypes.data_as(C.c_void_p), out.nbytes)
        capi.check(n)
        assert n == out.nbytes, (n, out.nbytes)
        return out

    def frames_seen(self):
        return capi.lib().bgs_group_frames_seen(self._h)

    def enable_kernel_timing(self, on=True):
        capi.check(capi.lib().bgs_group_enable_kernel_timing(self._h, 1 if on else 0))

    def kernel_timing(self):
        ms, n = C.c_double(0), C.c_int64(0)
        capi.check(capi.lib().bgs_group_kernel_timing(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value
