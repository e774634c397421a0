#!/usr/bin/env python3
"""Bulk kernels with unaligned ASCII pointers (input of encode / scan / k-mer batch, output of decode) vs aligned."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream)
n = 10**9
bufs = [torch.empty(n + 64, dtype=torch.uint8, device=dev) for _ in range(2)]
outs = [torch.empty(n + 64, dtype=torch.uint8, device=dev) for _ in range(2)]
words = [torch.empty((n + 31) // 32 + 2, dtype=torch.int64, device=dev) for _ in range(2)]
for b in bufs:
    ctx.nucgen_dev(b, n + 64, 5)
ctx.sync()


def timed(fn, reps=8):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record(stream)
    for i in range(reps):
        fn(i)
        ev[i + 1].record(stream)
    torch.cuda.synchronize()
    return statistics.median(ev[i].elapsed_time(ev[i + 1]) for i in range(2, reps))


nw = (n + 31) // 32
for off in (0, 1, 4, 8):
    e = timed(lambda i: ctx.encode_dev(bufs[i & 1].data_ptr() + off, n, words[i & 1]))
    d = timed(lambda i: ctx.decode_dev(words[i & 1], nw, n, outs[i & 1].data_ptr() + off))
    s = timed(lambda i: ctx.kmer_hdist_scan_dev(bufs[i & 1].data_ptr() + off, n, 31, 0x1B1B1B1B1B1B1B1B & ((1 << 62) - 1), outs[i & 1]))
    k = timed(lambda i: ctx.as_2bit_batch_dev(bufs[i & 1].data_ptr() + off, 31, 31, n // 31 // 4, words[i & 1]))
    ctx.sync()
    print(f"ASCII pointer offset {off}: encode {e:.4f} ms ({1.25 * n / e / 1e6:.0f} GB/s) | decode {d:.4f} ms ({1.25 * n / d / 1e6:.0f} GB/s) | "
          f"scan {s:.4f} ms ({2 * n / s / 1e6:.0f} GB/s) | dense 31-mers {k:.4f} ms ({39 * (n // 31 // 4) / k / 1e6:.0f} GB/s)", flush=True)
