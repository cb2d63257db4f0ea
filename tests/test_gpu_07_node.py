"""libbgs_node on the one GPU a test box has (include/bgs_node.h): the RCCL calls themselves through a communicator of one rank
(BGS_NODE_LOOPBACK: the root's block is sent to itself and received from itself), the rank form (ncclCommInitRank with a unique id),
and - with the same HIP device listed twice and peer copies instead of RCCL, which refuses two ranks on one device - everything else
of the N > 1 path: one host thread and two streams per device, ragged stream blocks, double-buffered gather, the host path routed to
the owning engine.  No N > 1 RCCL run exists on this pool (one GPU per box): these tests are the evidence there is."""
import numpy as np
import pytest

from gpu_helpers import _torch
from oracle import pyoracle
from tools import synth
from tracking_amd import capi, node

pytestmark = pytest.mark.gpu


def _unpack(words, n, H, W):
    raw = words.cpu().numpy().view(np.uint8).reshape(words.shape[0], -1)
    bits = np.unpackbits(raw, axis=1, bitorder="little")
    assert not bits[:, n:].any()
    return bits[:, :n].reshape(-1, H, W)


def _run(nd, devices_frames, clips, algo, H, W, T):
    """T steps; after each one the gathered masks (global stream order) must equal every stream's own oracle."""
    torch = _torch()
    S = clips.shape[0]
    orcs = [pyoracle.Oracle(algo) for _ in range(S)]
    out = torch.empty((S, nd.words_per_stream), dtype=torch.int64, device="cuda")
    for t in range(T):
        frames = [torch.from_numpy(np.ascontiguousarray(clips[f:f + c, t])).cuda() if c else None for f, c in devices_frames]
        flags = nd.step_device(frames)
        nd.copy_masks(out)  # ordered behind the gather on torch's stream; the next step is posted without waiting (double buffering)
        torch.cuda.synchronize()
        got = _unpack(out, H * W, H, W)
        for s in range(S):
            ofg, _ = orcs[s].process(clips[s, t], want_bg=False)
            assert bool(flags & capi.FG_VALID) == (ofg is not None)
            if ofg is not None:
                assert np.array_equal(got[s] * 255, np.where(ofg != 0, 255, 0)), (t, s)
    nd.sync()


@pytest.mark.parametrize("shape", [(48, 64), (37, 53)])
def test_one_rank_rccl_loopback_gather_equals_kernel_output(shape):
    """ncclCommInitAll over one device; every step the root's block goes out through ncclSend and comes back through ncclRecv."""
    H, W = shape
    S, T = 3, 5
    clips = np.stack([synth.random_frames(T, H, W, 3, seed=70 + s) for s in range(S)])
    nd = node.Node(capi.MOG2, S, devices=[0], transport=node.RCCL, flags=node.LOOPBACK)
    nd.set_geometry(H, W, 3)
    assert nd.is_root and nd.local_devices == 1 and nd.words_per_stream == (H * W + 63) // 64
    _run(nd, [(0, S)], clips, capi.MOG2, H, W, T)
    nd.close()


def test_rank_form_with_unique_id_world_of_one():
    """bgs_node_unique_id + bgs_node_create_rank (ncclCommInitRank), loopback; and without loopback: the root writes in place, no RCCL call."""
    H, W, S, T = 40, 72, 2, 4
    clips = np.stack([synth.random_frames(T, H, W, 3, seed=90 + s) for s in range(S)])
    uid = node.unique_id()
    assert len(uid) == node.ID_BYTES
    for flags in (node.LOOPBACK, 0):
        nd = node.Node.rank(capi.FRAME_DIFF, S, device=0, rank=0, world=1, root_rank=0, uid=uid if flags else None, flags=flags)
        nd.set_geometry(H, W, 3)
        _run(nd, [(0, S)], clips, capi.FRAME_DIFF, H, W, T)
        nd.close()


@pytest.mark.parametrize("algo", [capi.MOG2, capi.SUBSENSE])
@pytest.mark.parametrize("root", [0, 1])
def test_two_device_threads_ragged_blocks_peer_copy(algo, root):
    """Five cameras over two 'devices' (3 + 2; both are HIP device 0): two worker threads, two engines, four streams, the gather as
    peer copies into the root's buffer, either device as root."""
    H, W, S, T = 40, 72, 5, 6
    clips = np.stack([synth.random_frames(T, H, W, 3, seed=30 + s) for s in range(S)])
    nd = node.Node(algo, S, devices=[0, 0], root_index=root, transport=node.PEER_COPY, flags=node.ALLOW_DUPLICATE_DEVICES)
    nd.set_geometry(H, W, 3)
    assert nd.local_devices == 2
    blocks = [nd.local_block(i)[1:] for i in range(2)]
    assert blocks == [(0, 3), (3, 2)]
    _run(nd, blocks, clips, algo, H, W, T)
    ms, n = nd.step_stats()
    assert n == T and ms > 0
    nd.close()


def test_more_devices_than_streams_and_duplicate_devices_refused_for_rccl():
    H, W, T = 32, 64, 3
    clips = np.stack([synth.random_frames(T, H, W, 3, seed=5)])
    nd = node.Node(capi.ABL, 1, devices=[0, 0, 0], root_index=2, transport=node.PEER_COPY, flags=node.ALLOW_DUPLICATE_DEVICES)  # devices 1 and 2 own nothing
    nd.set_geometry(H, W, 3)
    _run(nd, [nd.local_block(i)[1:] for i in range(3)], clips, capi.ABL, H, W, T)
    nd.close()
    with pytest.raises(node.NodeError) as ei:
        node.Node(capi.MOG2, 4, devices=[0, 0], transport=node.RCCL)
    assert "listed twice" in str(ei.value)


def test_host_path_routed_to_the_owning_engine():
    """bgs_node_process(global stream): IBGS::process for a camera, wherever its model lives."""
    H, W, S, T = 36, 60, 4, 5
    clips = np.stack([synth.random_frames(T, H, W, 3, seed=120 + s) for s in range(S)])
    nd = node.Node(capi.MOG2, S, devices=[0, 0], transport=node.PEER_COPY, flags=node.ALLOW_DUPLICATE_DEVICES)
    orcs = [pyoracle.Oracle(capi.MOG2) for _ in range(S)]
    for t in range(T):
        for s in (2, 0, 3, 1):
            fg, bg = nd.process(clips[s, t], s, want_bg=True)
            ofg, obg = orcs[s].process(clips[s, t])
            assert np.array_equal(fg, ofg) and np.array_equal(bg, obg), (t, s)
    with pytest.raises(node.NodeError):
        nd.process(clips[0, 0], S)
    nd.close()
