#!/usr/bin/env python3
"""Strided (non-dense) k-mer batch: kmer_batch_kernel vs encode_fixed_kernel on the same layout."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream)
count = 10**8 // 2
seq = torch.empty(count * 64 + 64, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(seq, seq.numel(), 3)
out = torch.empty(count, dtype=torch.int64, device=dev)
out2 = torch.empty(count, dtype=torch.int64, device=dev)
ctx.sync()


def timed(fn, reps=8):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record(stream)
    for i in range(reps):
        fn()
        ev[i + 1].record(stream)
    torch.cuda.synchronize()
    return statistics.median(ev[i].elapsed_time(ev[i + 1]) for i in range(2, reps))


# every window of a sequence (`for w in seq.windows(k) { as_2bit(w) }`, src/lib.rs:170-173): stride 1, 9 B per window
nwin = 5 * 10**8
wout = torch.empty(nwin, dtype=torch.int64, device=dev)
for k in (31, 21, 4):
    a = timed(lambda: ctx.as_2bit_batch_dev(seq, k, 1, nwin, wout))
    ctx.sync()
    print(f"k={k} stride=1 ({nwin} windows): as_2bit_batch {a:.4f} ms {nwin * 9 / a / 1e6:6.0f} GB/s (1 B read + 8 B written per window)", flush=True)
del wout

for k, stride in [(31, 31), (31, 32), (21, 22), (32, 33), (31, 64), (16, 17)]:
    a = timed(lambda: ctx.as_2bit_batch_dev(seq, k, stride, count, out))
    b = timed(lambda: ctx.encode_fixed_dev(seq, k, stride, count, out2))
    ctx.sync()
    same = bool(torch.equal(out, out2))
    alg = count * (k + 8)
    print(f"k={k} stride={stride}: as_2bit_batch {a:.4f} ms {alg/a/1e6:6.0f} GB/s | encode_fixed {b:.4f} ms {alg/b/1e6:6.0f} GB/s | same={same}", flush=True)
