"""ctypes binding of the CPU oracle (oracle/libbgs_oracle.so) and of the reference-built checkers in
oracle/_ref/.  TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never by anything under tracking_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("BGS_ORACLE_LIB") or os.path.join(_HERE, "libbgs_oracle.so")  # BGS_ORACLE_LIB: e.g. an ASan/UBSan build (tools/sanitize_cpu.sh)
_REF_LBSP = os.path.join(_HERE, "_ref", "libref_lbsp.so")
_REF_SDLAMA = os.path.join(_HERE, "_ref", "ref_sdlama_cli")

_P = C.c_void_p
_lib = None


def build():
    """(Re)build the oracle; also builds oracle/_ref when /root/reference exists (this container only)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        l = C.CDLL(_LIB)
        l.orc_default_params.argtypes = [C.c_int, _P]
        l.orc_create.argtypes = [C.c_int, _P, C.POINTER(_P)]
        l.orc_set_params.argtypes = [_P, _P]
        l.orc_set_threads.argtypes = [_P, C.c_int]
        l.orc_process.argtypes = [_P, _P, C.c_int, C.c_int, C.c_int, C.c_size_t, _P, C.c_size_t, _P, C.c_size_t, C.POINTER(C.c_uint32)]
        l.orc_get_state.argtypes = [_P, C.c_char_p, _P, C.c_size_t]
        l.orc_get_state.restype = C.c_int64
        l.orc_destroy.argtypes = [_P]
        l.orc_destroy.restype = None
        l.orc_bgr2gray.argtypes = [_P, C.c_size_t, _P, C.c_size_t, C.c_int, C.c_int]
        l.orc_bgr2gray.restype = None
        l.orc_lbsp_lut.argtypes = [C.c_float, C.c_int, C.c_int, _P]
        l.orc_lbsp_lut.restype = None
        l.orc_lbsp_describe.argtypes = [_P, C.c_size_t, C.c_int, C.c_int, C.c_int, _P, _P]
        l.orc_lbsp_describe.restype = None
        for n in ("orc_median_blur_u8",):
            getattr(l, n).argtypes = [_P, _P, C.c_int, C.c_int, C.c_int]
            getattr(l, n).restype = None
        for n in ("orc_erode3x3", "orc_dilate3x3"):
            getattr(l, n).argtypes = [_P, _P, C.c_int, C.c_int, C.c_int]
            getattr(l, n).restype = None
        l.orc_floodfill_from_origin.argtypes = [_P, C.c_int, C.c_int, C.c_uint8]
        l.orc_floodfill_from_origin.restype = None
        l.orc_components.argtypes = [_P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int]
        l.orc_components.restype = C.c_int
        l.orc_ingest_size.argtypes = [_P, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        l.orc_ingest_size.restype = None
        l.orc_ingest.argtypes = [_P, _P, C.c_int, C.c_int, C.c_int, C.c_size_t, _P]
        l.orc_resize_linear_u8.argtypes = [_P, C.c_int, C.c_int, C.c_int, C.c_size_t, _P, C.c_int, C.c_int]
        l.orc_resize_linear_u8.restype = None
        l.orc_equalize_hist_u8.argtypes = [_P, C.c_size_t]
        l.orc_equalize_hist_u8.restype = None
        l.orc_gaussian7_kernel.argtypes = [_P]
        l.orc_resize_area_u8.argtypes = [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int]
        l.orc_resize_area_u8.restype = None
        l.orc_gaussian_blur7_u8.argtypes = [_P, _P, C.c_int, C.c_int, C.c_int]
        l.orc_gaussian_blur7_u8.restype = None
        _lib = l
    return _lib


def _ptr(a):
    return a.ctypes.data_as(_P) if a is not None else None


class Oracle:
    """CPU twin of tracking_amd.Engine.process for one stream."""

    def __init__(self, algo, params=None, threads=1):
        from tracking_amd import capi  # struct layout only (interface), no product code runs
        self._capi = capi
        self.algo = algo
        self._h = _P()
        if params is None:
            params = capi.BgsParams()
            params.struct_size = C.sizeof(capi.BgsParams)
            assert lib().orc_default_params(algo, C.byref(params)) == 0
        self.params = params
        rc = lib().orc_create(algo, C.byref(params), C.byref(self._h))
        assert rc == 0, rc
        if threads > 1:
            lib().orc_set_threads(self._h, threads)

    def set_params(self, params):
        assert lib().orc_set_params(self._h, C.byref(params)) == 0
        self.params = params

    def close(self):
        if self._h and self._h.value:
            lib().orc_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def process(self, frame, want_bg=True):
        if frame is None or frame.size == 0:
            flags = C.c_uint32(0)
            rc = lib().orc_process(self._h, None, 0, 0, 3, 0, None, 0, None, 0, C.byref(flags))
            assert rc == 0
            return None, None
        rows, cols = frame.shape[:2]
        ch = 1 if frame.ndim == 2 else frame.shape[2]
        if frame.strides[-1] != 1 or (frame.ndim == 3 and frame.strides[1] != ch):
            frame = np.ascontiguousarray(frame)
        fg = np.empty((rows, cols), np.uint8)
        bg_ch = 1 if self.algo == self._capi.ASBL else ch
        bg = np.empty((rows, cols, bg_ch), np.uint8) if want_bg else None
        flags = C.c_uint32(0)
        rc = lib().orc_process(self._h, _ptr(frame), rows, cols, ch, frame.strides[0], _ptr(fg), cols, _ptr(bg), cols * bg_ch, C.byref(flags))
        if rc != 0:
            raise RuntimeError("oracle error %d" % rc)
        f = flags.value
        if bg is not None and bg_ch == 1:
            bg = bg[:, :, 0]
        return (fg if f & 1 else None), (bg if (bg is not None and f & 2) else None)

    def get_state(self, plane, shape, dtype):
        out = np.empty(shape, dtype)
        n = lib().orc_get_state(self._h, plane.encode(), _ptr(out), out.nbytes)
        assert n == out.nbytes, (plane, n, out.nbytes)
        return out


def bgr2gray(img):
    rows, cols = img.shape[:2]
    img = np.ascontiguousarray(img)
    out = np.empty((rows, cols), np.uint8)
    lib().orc_bgr2gray(_ptr(img), img.strides[0], _ptr(out), cols, rows, cols)
    return out


def lbsp_lut(rel=0.333, offset=0, channels=3):
    lut = np.empty(256, np.uint8)
    lib().orc_lbsp_lut(C.c_float(rel), offset, channels, _ptr(lut))
    return lut


def lbsp_describe(img, lut):
    img = np.ascontiguousarray(img)
    rows, cols = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty((rows, cols, ch), np.uint16)
    lib().orc_lbsp_describe(_ptr(img), img.strides[0], rows, cols, ch, _ptr(lut), _ptr(out))
    return out


def median_blur(img, k):
    img = np.ascontiguousarray(img)
    out = np.empty_like(img)
    lib().orc_median_blur_u8(_ptr(img), _ptr(out), img.shape[0], img.shape[1], k)
    return out


def erode3x3(img, iterations=1):
    img = np.ascontiguousarray(img)
    out = np.empty_like(img)
    lib().orc_erode3x3(_ptr(img), _ptr(out), img.shape[0], img.shape[1], iterations)
    return out


def dilate3x3(img, iterations=1):
    img = np.ascontiguousarray(img)
    out = np.empty_like(img)
    lib().orc_dilate3x3(_ptr(img), _ptr(out), img.shape[0], img.shape[1], iterations)
    return out


def resize_area(img, drows, dcols):
    """cv::resize(img, (dcols, drows), INTER_AREA), general path (oracle/subsense_oracle.c area_value)."""
    img = np.ascontiguousarray(img, np.uint8)
    ch = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty((drows, dcols) if img.ndim == 2 else (drows, dcols, ch), np.uint8)
    lib().orc_resize_area_u8(_ptr(img), img.shape[0], img.shape[1], ch, _ptr(out), drows, dcols)
    return out


def floodfill_from_origin(img, newval=255):
    out = np.ascontiguousarray(img).copy()
    lib().orc_floodfill_from_origin(_ptr(out), out.shape[0], out.shape[1], newval)
    return out


BOX_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4"), ("area", "<i4"), ("root", "<i4")])


def components(mask, connectivity=8, max_boxes=1 << 20):
    """(labels int32 [rows][cols] = root index or -1, boxes[count] as BOX_DTYPE sorted by root)."""
    m = np.ascontiguousarray(mask, np.uint8)
    labels = np.empty(m.shape, np.int32)
    boxes = np.zeros(max_boxes, BOX_DTYPE)
    n = lib().orc_components(_ptr(m), m.shape[0], m.shape[1], connectivity, _ptr(labels), _ptr(boxes), max_boxes)
    return labels, boxes[:min(n, max_boxes)], n


# ---- reference-built checkers (oracle/_ref, compiled from /root/reference sources in this container) ----

def ref_lbsp_available():
    return os.path.exists(_REF_LBSP)


def ref_lbsp_describe(img, lut):
    """The reference's own LBSP_16bits_dbcross_{3ch3t,1ch}.i applied to every interior pixel."""
    l = C.CDLL(_REF_LBSP)
    l.ref_lbsp_describe.argtypes = [_P, C.c_size_t, C.c_int, C.c_int, C.c_int, _P, _P]
    l.ref_lbsp_describe.restype = None
    img = np.ascontiguousarray(img)
    rows, cols = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty((rows, cols, ch), np.uint16)
    l.ref_lbsp_describe(_ptr(img), img.strides[0], rows, cols, ch, _ptr(lut), _ptr(out))
    return out


def ref_sdlama_available():
    return os.path.exists(_REF_SDLAMA)


def ref_sigmadelta_clip(frames, amp=1, vmin=15, vmax=255):
    """Masks of frames[1:] from the REFERENCE'S OWN package_bgs/bl/sdLaMa091.cpp (compiled as is into oracle/_ref/ref_sdlama_cli)
    driven the way SigmaDeltaBGS::process drives it (SigmaDeltaBGS.cpp:20-55).  Runs in a process of its own so that the
    part of Vt sdLaMa091 never initialises is untouched zero memory (see oracle/ref_sdlama_cli.cpp)."""
    import tempfile
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    n, rows, cols = frames.shape[:3]
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.raw"), os.path.join(d, "out.raw")
        frames.tofile(fin)
        subprocess.run([_REF_SDLAMA, fin, str(rows), str(cols), str(n), str(amp), str(vmin), str(vmax), fout], check=True)
        return np.fromfile(fout, np.uint8).reshape(n - 1, rows, cols)


# ---- N3 frame preparation (ingest_oracle.c) ----

def ingest(cfg, frame):
    """cfg: tracking_amd.capi.BgsIngest.  frame: HxW or HxWxC uint8.  Returns the prepared frame (or None where the reference fails)."""
    f = np.ascontiguousarray(frame)
    rows, cols = f.shape[:2]
    ch = 1 if f.ndim == 2 else f.shape[2]
    r, c = C.c_int(0), C.c_int(0)
    lib().orc_ingest_size(C.byref(cfg), rows, cols, C.byref(r), C.byref(c))
    if r.value < 1 or c.value < 1:
        return None
    out = np.empty((r.value, c.value) if f.ndim == 2 else (r.value, c.value, ch), np.uint8)
    rc = lib().orc_ingest(C.byref(cfg), _ptr(f), rows, cols, ch, f.strides[0], _ptr(out))
    return out if rc == 0 else None


def gaussian7_kernel():
    ik = (C.c_int * 7)()
    smooth = lib().orc_gaussian7_kernel(ik)
    return list(ik), bool(smooth)
