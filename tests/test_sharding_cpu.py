"""N > 1 path on CPU: world_size-2 `gloo` run of the stream sharding + packed-mask gather (tracking_amd/sharding.py).
The per-rank masks come from the CPU oracle here (no GPU in this container); on the GPU box bench.py feeds the same
MaskGather with the kernel's bit-packed output."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tracking_amd.sharding import MaskGather, owner_of, stream_block

H, W, T, TOTAL = 16, 64, 6, 5  # 5 streams over 2 ranks: blocks of 3 and 2 (ragged on purpose)


def _masks_for_stream(s):
    from oracle import pyoracle
    from tools import synth
    from tracking_amd import capi
    frames = synth.random_frames(T, H, W, 3, seed=500 + s)
    o = pyoracle.Oracle(capi.MOG2)
    out = []
    for f in frames:
        fg, _ = o.process(f, want_bg=False)
        out.append(np.packbits((fg != 0).reshape(-1), bitorder="little").view(np.int64))
    return np.stack(out)  # [T][H*W/64]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        first, count = stream_block(TOTAL, world, rank)
        # every rank gathers the same shape: pad the ragged block to the largest one
        cap = max(stream_block(TOTAL, world, r)[1] for r in range(world))
        local = np.zeros((cap, T, H * W // 64), np.int64)
        for i in range(count):
            local[i] = _masks_for_stream(first + i)
        g = MaskGather(cap, H * W // 64, "cpu", dst=0)
        got = []
        for t in range(T):
            buf = g.next_buffer()
            buf.copy_(torch.from_numpy(local[:, t]))
            g.post()
            if t >= 1 and rank == 0:  # overlap: collect step t-1 ... here simply collect the step just posted
                pass
            res = g.collect()
            if rank == 0:
                got.append(res.clone().numpy().reshape(world, cap, -1))
        g.drain()
        if rank == 0:
            q.put(np.stack(got))
    finally:
        dist.destroy_process_group()


def _worker_overlapped(rank, world, port, q):
    """The order bench.py runs at N > 1: post(t) -> fill the OTHER buffer with step t+1 (the next kernel) -> collect(t)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        first, count = stream_block(TOTAL, world, rank)
        cap = max(stream_block(TOTAL, world, r)[1] for r in range(world))
        local = np.zeros((cap, T, H * W // 64), np.int64)
        for i in range(count):
            local[i] = _masks_for_stream(first + i)
        g = MaskGather(cap, H * W // 64, "cpu", dst=0)
        got = []
        g.next_buffer().copy_(torch.from_numpy(local[:, 0]))
        for t in range(T):
            g.post()                                   # gather of step t in flight
            if t + 1 < T:
                nxt = g.next_buffer()                  # the other buffer: step t+1's "kernel" writes it while step t travels
                assert nxt.data_ptr() != g.bufs[g.last].data_ptr()
                nxt.copy_(torch.from_numpy(local[:, t + 1]))
            res = g.collect()                          # step t, complete and in global stream order
            if rank == 0:
                got.append(res.clone().numpy().reshape(world, cap, -1))
            else:
                assert res is None
        g.drain()
        if rank == 0:
            q.put(np.stack(got))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_stream_blocks_partition_all_streams():
    for total in (1, 5, 32, 256, 257):
        for world in (1, 2, 4, 8):
            seen = []
            for r in range(world):
                f, c = stream_block(total, world, r)
                seen += list(range(f, f + c))
            assert seen == list(range(total))
            for s in (0, total - 1, total // 2):
                f, c = stream_block(total, world, owner_of(s, total, world))
                assert f <= s < f + c
    assert stream_block(256, 8, 3) == (96, 32)  # BASELINE config 5: 32 streams per GPU, contiguous


def test_two_rank_gloo_mask_gather_matches_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)  # [T][world][cap][words]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for s in range(TOTAL):
        r = owner_of(s, TOTAL, world)
        first, _ = stream_block(TOTAL, world, r)
        want = _masks_for_stream(s)  # [T][words]
        assert np.array_equal(got[:, r, s - first], want), "stream %d" % s


@pytest.mark.parametrize("world", [2, 3])
def test_overlapped_gather_keeps_per_stream_order(world):
    """post(t) / fill(t+1) / collect(t) over 6 steps with ragged stream blocks (5 streams over 2 or 3 ranks): what rank 0 collects
    at step t is every stream's mask of step t - never the step the next kernel is writing."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_overlapped, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)  # [T][world][cap][words]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got.shape[0] == T
    for s in range(TOTAL):
        r = owner_of(s, TOTAL, world)
        first, _ = stream_block(TOTAL, world, r)
        want = _masks_for_stream(s)
        for t in range(T):
            assert np.array_equal(got[t, r, s - first], want[t]), "stream %d step %d" % (s, t)


def test_single_rank_gather_is_a_passthrough():
    g = MaskGather(3, 4, "cpu")
    b = g.next_buffer()
    b.fill_(7)
    g.post()
    assert torch.equal(g.collect(), torch.full((3, 4), 7, dtype=torch.int64))
