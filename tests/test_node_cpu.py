"""libbgs_node (include/bgs_node.h) without a GPU: the library loads, exports and binds every symbol its header declares, the
stream -> device map equals the one the torch.distributed path uses (tracking_amd/sharding.py), and creation fails loudly."""
import ctypes as C
import os
import re

import pytest

from tracking_amd import capi, node
from tracking_amd.sharding import owner_of, stream_block

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bgs_node.h")


def test_every_declared_node_symbol_is_exported_and_bound():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(bgs_node_[a-z0-9_]+)\s*\(", text)))
    assert len(names) >= 20
    lib = C.CDLL(node.LIB_PATH)
    assert not [n for n in names if not hasattr(lib, n)]
    assert sorted(n for n, _, _ in node.SYMBOLS) == names, "node.SYMBOLS must bind exactly what the header declares"


@pytest.mark.parametrize("total,n", [(256, 8), (32, 1), (5, 2), (11, 3), (7, 8), (1, 4), (100, 7)])
def test_stream_map_equals_the_sharding_module(total, n):
    """bgs_node_stream_block / _owner == stream_block / owner_of: contiguous blocks, the first total % n devices one stream more."""
    covered = []
    for i in range(n):
        f, c = node.stream_block(total, n, i)
        assert (f, c) == stream_block(total, n, i)
        covered += list(range(f, f + c))
    assert covered == list(range(total))
    for s in range(total):
        idx, local = node.stream_owner(total, n, s)
        assert idx == owner_of(s, total, n)
        assert local == s - stream_block(total, n, idx)[0]


def test_bad_arguments_and_no_gpu_are_reported():
    l = node.lib()
    h = C.c_void_p()
    assert l.bgs_node_stream_block(4, 0, 0, None, None) == capi.ERR_INVALID
    assert l.bgs_node_stream_owner(4, 2, 9, None, None) == capi.ERR_INVALID
    assert l.bgs_node_create(99, None, None, 1, 4, 0, node.RCCL, 0, C.byref(h)) == capi.ERR_INVALID
    assert l.bgs_node_create(capi.MOG2, None, None, 2, 4, 5, node.RCCL, 0, C.byref(h)) == capi.ERR_INVALID
    assert b"root_index" in l.bgs_node_last_error()
    import torch
    if not torch.cuda.is_available():
        rc = l.bgs_node_create(capi.MOG2, None, None, 1, 4, 0, node.RCCL, 0, C.byref(h))
        assert rc == capi.ERR_HIP and not h.value
        assert b"no CPU path" in l.bgs_node_last_error()
