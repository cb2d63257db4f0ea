import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip silently: the product has no CPU path.
    pass


@pytest.fixture(scope="session")
def golden_frames():
    return np.load(os.path.join(GOLDEN, "frames_96x80.npz"))["frames"]


@pytest.fixture(scope="session")
def golden_gray():
    return np.load(os.path.join(GOLDEN, "frames_gray_64x48.npz"))["frames"]


@pytest.fixture(scope="session")
def oracle_regress():
    return np.load(os.path.join(GOLDEN, "oracle_regress.npz"))


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the oracle (and the HIP library if it is missing) once per session."""
    from oracle import pyoracle
    pyoracle.lib()
    from tracking_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__ as g
        g.build()
