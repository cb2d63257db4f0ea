"""ctypes binding of libbgs_hip's C ABI (include/bgs_hip.h) — plumbing only.

This module never computes anything itself and has no CPU path: if the HIP library is
missing it raises at import of the symbols, and every call into a box without a GPU
fails with BgsError(BGS_ERR_HIP).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BGS_LIB_PATH") or os.path.join(_HERE, "lib", "libbgs_hip.so")  # BGS_LIB_PATH: A/B builds of the same ABI (tools/)

# bgs_algo (include/bgs_hip.h)
FRAME_DIFF, STATIC_FRAME_DIFF, WMM, WMV, ABL, ASBL, MOG2, MOG1, GMG, SUBSENSE, LBSP_DESC, SIGMA_DELTA = range(12)
DP_ZIVKOVIC_AGMM, DP_GRIMSON_GMM, DP_WREN_GA, DP_MEAN, DP_ADAPTIVE_MEDIAN = range(12, 17)
LOBSTER = 17
FG_VALID, BG_VALID = 1, 2
OPT_BORROW_FRAMES, OPT_MOG2_PIXELS_PER_LANE, OPT_MOG2_TILED, OPT_XCD_SWIZZLE, OPT_PLACEMENT_PROBE, OPT_MOG2_SPARSE, OPT_CLIP_FUSE, OPT_HOST_REGISTER, OPT_MODEL_CHUNK_MB, OPT_MODEL_CHUNK_MIN_MB = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10

OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_GEOMETRY, ERR_HIP, ERR_NOMEM, ERR_STATE = 0, -1, -2, -3, -4, -5, -6


class BgsParams(C.Structure):
    """struct bgs_params, field for field."""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("enable_threshold", C.c_int32),
        ("threshold", C.c_int32),
        ("enable_weight", C.c_int32),
        ("alpha", C.c_double),
        ("limit", C.c_int32),
        ("learning_frames", C.c_int32),
        ("alpha_learn", C.c_double),
        ("alpha_detection", C.c_double),
        ("mog2_history", C.c_int32),
        ("mog2_nmixtures", C.c_int32),
        ("mog2_var_threshold", C.c_float),
        ("mog2_background_ratio", C.c_float),
        ("mog2_var_threshold_gen", C.c_float),
        ("mog2_var_init", C.c_float),
        ("mog2_var_min", C.c_float),
        ("mog2_var_max", C.c_float),
        ("mog2_ct", C.c_float),
        ("mog2_tau", C.c_float),
        ("mog2_detect_shadows", C.c_int32),
        ("mog2_shadow_value", C.c_int32),
        ("mog1_history", C.c_int32),
        ("mog1_nmixtures", C.c_int32),
        ("mog1_background_ratio", C.c_double),
        ("mog1_var_threshold", C.c_double),
        ("mog1_noise_sigma", C.c_double),
        ("lbsp_rel_threshold", C.c_float),
        ("lbsp_threshold_offset", C.c_int32),
        ("subsense_min_color_dist_threshold", C.c_int32),
        ("subsense_n_samples", C.c_int32),
        ("subsense_n_required", C.c_int32),
        ("subsense_samples_for_moving_avgs", C.c_int32),
        ("sd_amp_factor", C.c_int32),
        ("sd_min_var", C.c_int32),
        ("sd_max_var", C.c_int32),
        ("subsense_desc_dist_threshold_offset", C.c_int32),
        ("gmg_max_features", C.c_int32),
        ("gmg_init_frames", C.c_int32),
        ("gmg_quantization_levels", C.c_int32),
        ("gmg_smoothing_radius", C.c_int32),
        ("gmg_update_background_model", C.c_int32),
        ("dp_sampling_rate", C.c_int32),
        ("gmg_learning_rate", C.c_double),
        ("gmg_background_prior", C.c_double),
        ("gmg_decision_threshold", C.c_double),
        ("dp_threshold", C.c_float),
        ("dp_alpha", C.c_float),
        ("dp_gaussians", C.c_int32),
    ]


class BgsError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("libbgs_hip error %d: %s" % (code, text))
        self.code = code


# every symbol include/bgs_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("bgs_abi_version", C.c_int, []),
    ("bgs_default_params", C.c_int, [C.c_int, C.POINTER(BgsParams)]),
    ("bgs_create", C.c_int, [C.c_int, C.POINTER(BgsParams), C.c_int, C.c_int, C.POINTER(_P)]),
    ("bgs_set_params", C.c_int, [_P, C.POINTER(BgsParams)]),
    ("bgs_set_option", C.c_int, [_P, C.c_int, C.c_int64]),
    ("bgs_set_geometry", C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    ("bgs_process", C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_size_t, _P, C.c_size_t, _P, C.c_size_t, C.POINTER(C.c_uint32)]),
    ("bgs_process_batch_device", C.c_int, [_P, _P, _P, _P, _P, _P, C.POINTER(C.c_uint32)]),
    ("bgs_process_range_device", C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.POINTER(C.c_uint32)]),
    ("bgs_process_clip_device", C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.POINTER(C.c_uint32)]),
    ("bgs_get_state", C.c_int64, [_P, C.c_int, C.c_char_p, _P, C.c_size_t]),
    ("bgs_frames_seen", C.c_int64, [_P, C.c_int]),
    ("bgs_enable_kernel_timing", C.c_int, [_P, C.c_int]),
    ("bgs_kernel_timing", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_char_p)]),
    ("bgs_kernel_timing_series", C.c_int64, [_P, C.POINTER(C.c_float), C.c_int64]),
    ("bgs_calibrate_copy", C.c_int, [C.c_int, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    ("bgs_calibrate_pcie", C.c_int, [C.c_int, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("bgs_destroy", None, [_P]),
    ("bgs_last_error", C.c_char_p, []),
    ("bgs_lbsp_describe_device", C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    ("bgs_lbsp_describe_batch_device", C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    ("bgs_mask_morph_device", C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    ("bgs_mask_components_workspace", C.c_size_t, [C.c_int, C.c_int]),
    ("bgs_mask_components_device", C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, _P]),
    ("bgs_mask_components_batch_workspace", C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    ("bgs_mask_components_batch_device", C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, _P]),
    ("bgs_mask_blobs_batch_device", C.c_int, [C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, _P]),
    ("bgs_ingest_default", C.c_int, [_P]),
    ("bgs_ingest_size", C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("bgs_ingest_workspace", C.c_size_t, [_P, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("bgs_ingest_device", C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, _P, _P, _P]),
    ("bgs_set_ingest", C.c_int, [_P, _P]),
    ("bgs_ingest_host", C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_size_t, _P, C.c_size_t]),
    ("bgs_group_create", C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    ("bgs_group_destroy", None, [_P]),
    ("bgs_group_size", C.c_int, [_P]),
    ("bgs_group_is_fused", C.c_int, [_P, C.c_int]),
    ("bgs_group_set_params", C.c_int, [_P, C.c_int, _P]),
    ("bgs_group_set_option", C.c_int, [_P, C.c_int, C.c_int64]),
    ("bgs_group_set_geometry", C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    ("bgs_group_process_batch_device", C.c_int, [_P, _P, _P, _P, _P, C.POINTER(C.c_uint32)]),
    ("bgs_group_process", C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_size_t, _P, _P, _P, _P, C.POINTER(C.c_uint32)]),
    ("bgs_group_get_state", C.c_int64, [_P, C.c_int, C.c_int, C.c_char_p, _P, C.c_size_t]),
    ("bgs_group_frames_seen", C.c_int64, [_P]),
    ("bgs_group_enable_kernel_timing", C.c_int, [_P, C.c_int]),
    ("bgs_group_kernel_timing", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    ("bgs_submit", C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_size_t, _P, C.c_size_t, _P, C.c_size_t]),
    ("bgs_wait", C.c_int, [_P, C.c_int, C.POINTER(C.c_uint32)]),
    ("bgs_host_arena", C.c_int, [_P, _P, C.c_size_t, C.c_int]),
    ("bgs_reset_stream", C.c_int, [_P, C.c_int]),
    ("bgs_stream_flags", C.c_int, [_P, C.c_int, C.POINTER(C.c_uint32)]),
    ("bgs_last_mask_blobs", C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.POINTER(C.c_int32)]),
]

_lib = None


def lib():
    """Load libbgs_hip.so (built in-tree by __graft_entry__.build()).  No fallback of any kind."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libbgs_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` (%s)" % LIB_PATH)
        # torch ships its own copy of the HIP runtime (same soname, libamdhip64.so.7).  Whichever copy a process loads first serves
        # both; loaded in the other order - this library first, torch afterwards - torch came up with "No HIP GPUs are available"
        # on the GPU boxes (round 3, a test run that touched the C ABI before its first torch.cuda call).  So where torch exists
        # (tests, bench.py) it is imported first; a C++ host never has it and is not affected.
        try:
            import torch  # noqa: F401
        except Exception:  # noqa: BLE001
            pass
        l = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            if os.environ.get("BGS_LIB_PATH") and os.environ.get("BGS_LIB_PARTIAL_ABI") and not hasattr(l, name):
                continue  # A/B against an older build of the library (tools/): entry points it lacks simply stay unbound
            fn = getattr(l, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def last_error():
    return (lib().bgs_last_error() or b"").decode("utf-8", "replace")


def check(rc):
    if rc < 0:
        raise BgsError(rc, last_error())
    return rc


class BgsIngest(C.Structure):
    """struct bgs_ingest, field for field (N3 frame preparation)."""
    _fields_ = [("struct_size", C.c_uint32), ("resize_percent", C.c_int32), ("flip", C.c_int32), ("roi_x0", C.c_int32), ("roi_y0", C.c_int32),
                ("roi_x1", C.c_int32), ("roi_y1", C.c_int32), ("equalize_hist", C.c_int32), ("gaussian_blur", C.c_int32)]


def default_ingest(**kw):
    c = BgsIngest()
    check(lib().bgs_ingest_default(C.byref(c)))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def default_params(algo):
    p = BgsParams()
    p.struct_size = C.sizeof(BgsParams)
    check(lib().bgs_default_params(algo, C.byref(p)))
    return p


def calibrate_copy(device, nbytes, chunk_mb, iters=5):
    """bgs_calibrate_copy: GB/s of a float4 copy over nbytes through a plain (chunk_mb = 0) or chunked allocation."""
    g = C.c_double(0)
    check(lib().bgs_calibrate_copy(device, nbytes, chunk_mb, iters, C.byref(g)))
    return g.value


def calibrate_pcie(device, nbytes, registered, iters=20):
    """bgs_calibrate_pcie: (H2D GB/s, D2H GB/s, hipHostRegister ms)"""
    u, d, r = C.c_double(0), C.c_double(0), C.c_double(0)
    check(lib().bgs_calibrate_pcie(device, nbytes, 1 if registered else 0, iters, C.byref(u), C.byref(d), C.byref(r)))
    return u.value, d.value, r.value
