#!/usr/bin/env python3
"""Rate vs footprint for the read-mostly kernels whose BASELINE-size input is only 250-500 MB (base_counts, bulk hdist, hdist_query):
is their distance from 8 TB/s the kernel's, or the price of starting and draining a grid every 40-75 us?  Sustained bursts of 8
launches on rotating buffers (footprint of one burst > 2 GB, so the 256 MiB Infinity Cache cannot serve it), median of 5."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream)
BURST = 8


def sustained(fn):
    ts = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(0)
        a.record(stream)
        for i in range(BURST):
            fn(i)
        b.record(stream)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / BURST)
    return statistics.median(ts[1:])


print(f"{'bases':>14s} {'MB read':>9s} | {'base_counts':>22s} | {'hdist (2 buffers)':>24s} | {'hdist_query (+1 B/word out)':>28s}")
for nb in (125_000_000, 250_000_000, 500_000_000, 10**9, 2 * 10**9, 4 * 10**9, 8 * 10**9):
    nw = nb // 32
    R = max(2, min(8, (3 << 30) // (nw * 8) + 1))  # rotate over > 3 GB
    A = [torch.randint(-2**62, 2**62, (nw,), dtype=torch.int64, device=dev) for _ in range(R)]
    cnt = torch.zeros(4, dtype=torch.int64, device=dev)
    res = torch.zeros(1, dtype=torch.int32, device=dev)
    dist = torch.empty(nw, dtype=torch.uint8, device=dev)
    t1 = sustained(lambda i: ctx.base_counts_dev(A[i % R], nw, nb, cnt))
    t2 = sustained(lambda i: ctx.hdist_dev(A[i % R], nw, A[(i + 1) % R], nw, nb, res))
    t3 = sustained(lambda i: ctx.hdist_query_dev(0x1234567, A[i % R], nw, 32, dist))
    ctx.sync()
    f = lambda byts, ms: f"{ms * 1e3:8.1f} us {byts / ms / 1e6:6.0f} GB/s {byts / ms / 1e6 / 8000:.3f}"
    print(f"{nb:14d} {nw * 8 / 1e6:9.0f} | {f(nw * 8, t1)} | {f(2 * nw * 8, t2)} | {f(nw * 9, t3)}", flush=True)
    del A, dist
    torch.cuda.empty_cache()
