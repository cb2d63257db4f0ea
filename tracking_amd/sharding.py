"""Stream-level data parallelism: camera streams are independent units, so they shard across GPUs
with NO data-path collective — stream s lives on rank s // streams_per_rank (contiguous blocks, so each rank's masks
form one buffer).  The only exchange step is the final gather of the bit-packed foreground masks to the rank that
feeds the blob detector (CvBlobDetector consumes the mask right after FG detection, ustc_src/trackingMain.cpp:166).

xGMI is point-to-point: a gather to root is 7 independent transfers over 7 distinct links (per-link bound at the root,
not ring bound), and bit-packing makes it 8x smaller than byte masks (1080p: 259 200 B per stream and frame).
torch.distributed here is plumbing: backend "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests.
"""
import time

import torch
import torch.distributed as dist


def packed_words(rows, cols):
    """64-bit words of one stream's bit-packed mask (include/bgs_hip.h: W = ceil(rows*cols / 64), tail bits zero)."""
    return (rows * cols + 63) // 64


def stream_block(total_streams, world_size, rank):
    """Contiguous block [first, first+count) of the global stream ids owned by `rank`."""
    base, extra = divmod(total_streams, world_size)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def owner_of(stream, total_streams, world_size):
    for r in range(world_size):
        f, c = stream_block(total_streams, world_size, r)
        if f <= stream < f + c:
            return r
    raise IndexError(stream)


class MaskGather:
    """Double-buffered asynchronous gather of per-rank packed masks to `dst`.

    post(t) starts the gather of this step's masks and returns immediately, so the transfer overlaps the next
    step's update kernel (which writes the other buffer); collect() waits and, on dst, returns the masks of all
    ranks in global stream order as one [total_streams][words] tensor."""

    def __init__(self, streams_local, words, device, dst=0, group=None, always_collective=False):
        # always_collective: issue the gather even in a world of one rank (bench.py --rccl-selftest: the RCCL calls of the N > 1 path
        # on a box with a single GPU)
        self.always = always_collective and dist.is_initialized()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.dst, self.group = dst, group
        self.shape = (streams_local, words)
        self.bufs = [torch.zeros(self.shape, dtype=torch.int64, device=device) for _ in range(2)]
        self.recv = [[torch.empty(self.shape, dtype=torch.int64, device=device) for _ in range(self.world)] if self.rank == dst else None for _ in range(2)]
        self.work = [None, None]
        self.i = 0
        self.blocked_s, self.blocked_calls = 0.0, 0  # host time next_buffer() spent waiting for an earlier gather (diagnostics of a scaling run)

    def reset_stats(self):
        self.blocked_s, self.blocked_calls = 0.0, 0

    def next_buffer(self):
        """Buffer the update kernel should write this step (waits for the gather that last used it)."""
        if self.work[self.i] is not None:
            t0 = time.perf_counter()
            self.work[self.i].wait()
            self.blocked_s += time.perf_counter() - t0
            self.work[self.i] = None
        self.blocked_calls += 1
        return self.bufs[self.i]

    def post(self):
        if self.world > 1 or self.always:
            self.work[self.i] = dist.gather(self.bufs[self.i], self.recv[self.i], dst=self.dst, group=self.group, async_op=True)
        self.last = self.i
        self.i ^= 1

    def collect(self):
        j = self.last
        if self.work[j] is not None:
            self.work[j].wait()
            self.work[j] = None
        if self.rank != self.dst:
            return None
        return torch.cat(self.recv[j], 0) if (self.world > 1 or self.always) else self.bufs[j]

    def drain(self):
        for j in (0, 1):
            if self.work[j] is not None:
                self.work[j].wait()
                self.work[j] = None
