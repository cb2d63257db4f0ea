#!/usr/bin/env python3
"""HBM calibration with plain torch ops (plumbing only): copy vs in-place read-modify-write at several sizes."""
import time
import torch

def bench(fn, bytes_moved, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return bytes_moved * n / (time.perf_counter() - t0) / 1e9

for gib in (1, 6):
    n = gib * 2**30 // 4
    a = torch.zeros(n, dtype=torch.float32, device="cuda")
    b = torch.zeros(n, dtype=torch.float32, device="cuda")
    print("%d GiB copy b<-a      : %7.1f GB/s" % (gib, bench(lambda: b.copy_(a), 8 * n)))
    print("%d GiB in-place a+=1  : %7.1f GB/s" % (gib, bench(lambda: a.add_(1.0), 8 * n)))
    print("%d GiB out-of-place b=a+1: %7.1f GB/s" % (gib, bench(lambda: torch.add(a, 1.0, out=b), 8 * n)))
    print("%d GiB read-only sum  : %7.1f GB/s" % (gib, bench(lambda: a.sum(), 4 * n)))
    print("%d GiB write-only fill: %7.1f GB/s" % (gib, bench(lambda: b.fill_(1.0), 4 * n)))
    del a, b
