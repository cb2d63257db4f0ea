"""tracking_amd — MI355X-native foreground detection behind the reference's IBGS plugin surface.

Product = tracking_amd/csrc (HIP kernels + C ABI, built into tracking_amd/lib/libbgs_hip.so) and
tracking_amd/host (C++ IBGS / FrameProcessor mirror above the C ABI).  The Python modules here are
bindings for tests and bench.py; they contain no arithmetic and no CPU fallback.
"""
from . import capi  # noqa: F401
from .engine import Engine  # noqa: F401
