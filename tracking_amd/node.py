"""ctypes binding of libbgs_node (include/bgs_node.h): one caller, all the GPUs of a node - plumbing only, for tests and bench.py.
The map, the threads, the streams and the RCCL calls live in tracking_amd/csrc/bgs_node.cpp."""
import ctypes as C
import os

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libbgs_node.so")
RCCL, PEER_COPY = 0, 1
LOOPBACK, ALLOW_DUPLICATE_DEVICES = 1, 2
ID_BYTES = 128

_P = C.c_void_p
_I = C.POINTER(C.c_int)
# every symbol include/bgs_node.h declares
SYMBOLS = [
    ("bgs_node_stream_block", C.c_int, [C.c_int, C.c_int, C.c_int, _I, _I]),
    ("bgs_node_stream_owner", C.c_int, [C.c_int, C.c_int, C.c_int, _I, _I]),
    ("bgs_node_create", C.c_int, [C.c_int, C.POINTER(capi.BgsParams), _I, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.POINTER(_P)]),
    ("bgs_node_unique_id", C.c_int, [_P]),
    ("bgs_node_create_rank", C.c_int, [C.c_int, C.POINTER(capi.BgsParams), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_uint, C.POINTER(_P)]),
    ("bgs_node_set_geometry", C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    ("bgs_node_set_option", C.c_int, [_P, C.c_int, C.c_int64]),
    ("bgs_node_local_devices", C.c_int, [_P]),
    ("bgs_node_local_block", C.c_int, [_P, C.c_int, _I, _I, _I]),
    ("bgs_node_engine", _P, [_P, C.c_int]),
    ("bgs_node_words_per_stream", C.c_size_t, [_P]),
    ("bgs_node_is_root", C.c_int, [_P]),
    ("bgs_node_step_device", C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint32)]),
    ("bgs_node_collect", C.c_int, [_P, C.POINTER(_P), _P]),
    ("bgs_node_copy_masks", C.c_int, [_P, _P, _P]),
    ("bgs_node_sync", C.c_int, [_P]),
    ("bgs_node_step_stats", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
    ("bgs_node_process", C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_size_t, _P, C.c_size_t, _P, C.c_size_t, C.POINTER(C.c_uint32)]),
    ("bgs_node_submit", C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_size_t, _P, C.c_size_t, _P, C.c_size_t]),
    ("bgs_node_wait", C.c_int, [_P, C.c_int, C.POINTER(C.c_uint32)]),
    ("bgs_node_destroy", None, [_P]),
    ("bgs_node_last_error", C.c_char_p, []),
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libbgs_node.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` (%s)" % LIB_PATH)
        capi.lib()  # libbgs_hip.so first (and torch before it where it exists: see capi.lib)
        l = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


class NodeError(RuntimeError):
    pass


def check(rc):
    if rc < 0:
        raise NodeError("libbgs_node error %d: %s" % (rc, (lib().bgs_node_last_error() or b"").decode("utf-8", "replace")))
    return rc


def stream_block(total, n_devices, index):
    f, c = C.c_int(0), C.c_int(0)
    check(lib().bgs_node_stream_block(total, n_devices, index, C.byref(f), C.byref(c)))
    return f.value, c.value


def stream_owner(total, n_devices, stream):
    i, l = C.c_int(0), C.c_int(0)
    check(lib().bgs_node_stream_owner(total, n_devices, stream, C.byref(i), C.byref(l)))
    return i.value, l.value


def unique_id():
    buf = C.create_string_buffer(ID_BYTES)
    check(lib().bgs_node_unique_id(buf))
    return buf.raw


class Node:
    """A bgs_node.  Single-process form: Node(algo, total_streams, devices=[0, 1, ...]); one process per GPU: Node.rank(...)."""

    def __init__(self, algo, total_streams, devices=(0,), root_index=0, transport=RCCL, flags=0, params=None, _handle=None):
        self._h = C.c_void_p()
        self.algo, self.total = algo, total_streams
        if _handle is not None:
            self._h = _handle
            return
        p = params if params is not None else capi.default_params(algo)
        devs = (C.c_int * len(devices))(*devices)
        check(lib().bgs_node_create(algo, C.byref(p), devs, len(devices), total_streams, root_index, transport, flags, C.byref(self._h)))

    @classmethod
    def rank(cls, algo, total_streams, device, rank, world, root_rank, uid, flags=0, params=None):
        h = C.c_void_p()
        p = params if params is not None else capi.default_params(algo)
        check(lib().bgs_node_create_rank(algo, C.byref(p), device, rank, world, total_streams, root_rank, uid, flags, C.byref(h)))
        return cls(algo, total_streams, _handle=h)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().bgs_node_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def set_geometry(self, rows, cols, channels):
        check(lib().bgs_node_set_geometry(self._h, rows, cols, channels))

    def set_option(self, option, value):
        check(lib().bgs_node_set_option(self._h, option, int(value)))

    @property
    def local_devices(self):
        return lib().bgs_node_local_devices(self._h)

    def local_block(self, i):
        d, f, c = C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib().bgs_node_local_block(self._h, i, C.byref(d), C.byref(f), C.byref(c)))
        return d.value, f.value, c.value

    def engine_handle(self, i):
        """the bgs_engine* of local device i (owned by the node), for capi calls such as bgs_get_state / bgs_enable_kernel_timing"""
        return C.c_void_p(lib().bgs_node_engine(self._h, i))

    @property
    def words_per_stream(self):
        return lib().bgs_node_words_per_stream(self._h)

    @property
    def is_root(self):
        return bool(lib().bgs_node_is_root(self._h))

    def step_device(self, frames):
        """frames: one torch CUDA tensor per local device ([count_i][rows][cols][ch] uint8, on that device).  Returns the out_flags."""
        ptrs = (C.c_void_p * len(frames))(*[(t.data_ptr() if t is not None else None) for t in frames])
        flags = C.c_uint32(0)
        check(lib().bgs_node_step_device(self._h, ptrs, C.byref(flags)))
        return flags.value

    def collect(self, hip_stream=None):
        """Waits for the gather of the last step; returns the device address of [total][words] uint64 on the root device (None elsewhere)."""
        p = C.c_void_p()
        check(lib().bgs_node_collect(self._h, C.byref(p), C.c_void_p(hip_stream) if hip_stream else None))
        return p.value

    def copy_masks(self, out, hip_stream=None):
        """bgs_node_copy_masks into `out` (torch int64 [total][words] on the root device), on hip_stream (default: torch's current)."""
        import torch
        if hip_stream is None:
            hip_stream = torch.cuda.current_stream(out.device).cuda_stream
        check(lib().bgs_node_copy_masks(self._h, C.c_void_p(out.data_ptr()), C.c_void_p(hip_stream)))
        return out

    def sync(self):
        check(lib().bgs_node_sync(self._h))

    def step_stats(self, reset=False):
        ms, n = C.c_double(0), C.c_int64(0)
        check(lib().bgs_node_step_stats(self._h, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    def process(self, frame, stream, want_bg=False):
        import numpy as np
        rows, cols = frame.shape[:2]
        ch = 1 if frame.ndim == 2 else frame.shape[2]
        frame = np.ascontiguousarray(frame)
        fg = np.empty((rows, cols), np.uint8)
        bg = np.empty((rows, cols, ch), np.uint8) if want_bg else None
        flags = C.c_uint32(0)
        check(lib().bgs_node_process(self._h, stream, frame.ctypes.data_as(C.c_void_p), rows, cols, ch, frame.strides[0], fg.ctypes.data_as(C.c_void_p), fg.strides[0],
                                     bg.ctypes.data_as(C.c_void_p) if bg is not None else None, bg.strides[0] if bg is not None else 0, C.byref(flags)))
        return (fg if flags.value & capi.FG_VALID else None), (bg if (bg is not None and flags.value & capi.BG_VALID) else None)
