#!/usr/bin/env python3
"""as_2bit_batch across (k, stride) shapes: time, k-mers/s, and GB/s of the bytes the batch spans + 8 B out per k-mer."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream)
N = 2 * 10**9
seq = torch.empty(N, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(seq, N, 3)
ctx.sync()


def timed(fn, reps=6):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record(stream)
    for i in range(reps):
        fn()
        ev[i + 1].record(stream)
    torch.cuda.synchronize()
    return statistics.median(ev[i].elapsed_time(ev[i + 1]) for i in range(2, reps))


for k in (31, 21):
    for stride in (1, 2, 3, 4, 8, 16, 24, k, 32, 33, 40, 64, 65, 100, 256):
        count = min(2 * 10**8, (N - k) // stride + 1)
        out = torch.empty(count, dtype=torch.int64, device=dev)
        ms = timed(lambda: ctx.as_2bit_batch_dev(seq, k, stride, count, out))
        ctx.sync()
        span = (count - 1) * stride + k            # bytes the batch spans (each read at most once)
        touched = count * min(stride, 128) if stride > k else span  # rough: cache lines touched when k-mers are sparse
        print(f"k={k} stride={stride:4d} count={count:.2e}: {ms:.4f} ms  {count / ms / 1e6:7.1f} G k-mers/s  "
              f"{(span + 8 * count) / ms / 1e6:6.0f} GB/s (span + out)", flush=True)
        del out
