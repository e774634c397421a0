#!/usr/bin/env python3
"""Resident-grid size (workgroups per CU) of the single-launch reductions: base_counts and bulk hdist, 10^9 bases,
inputs alternating between two packed buffers (cache-cold)."""
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
nw = n // 32
seq = torch.empty(n, dtype=torch.uint8, device=dev)
w = [torch.empty(nw, dtype=torch.int64, device=dev) for _ in range(4)]
for i in range(4):
    ctx.nucgen_dev(seq, n, 10 + i)
    ctx.encode_dev(seq, n, w[i])
ctx.sync()
cnt = torch.zeros(4, dtype=torch.int64, device=dev)
res = torch.zeros(1, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
rows = {}
for rnd in range(7):
    for mult in (1, 2, 3, 4, 8):
        ctx.require_variant("reduce_mult", mult)
        for name, fn in (("base_counts", lambda i: ctx.base_counts_dev(w[i & 1], nw, n, cnt)),
                         ("hdist", lambda i: ctx.hdist_dev(w[i & 1], nw, w[2 + (i & 1)], nw, n, res))):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
            ev[0].record(stream)
            for i in range(8):
                fn(i)
                ev[i + 1].record(stream)
            torch.cuda.synchronize()
            rows.setdefault((name, mult), []).append(statistics.mean(ev[i].elapsed_time(ev[i + 1]) for i in range(2, 8)))
ctx.sync()
for name, bytes_ in (("base_counts", 8 * nw), ("hdist", 16 * nw)):
    for mult in (1, 2, 3, 4, 8):
        ms = statistics.median(rows[(name, mult)])
        print(f"{name:12s} {mult:2d} workgroups/CU: {ms * 1e3:7.1f} us  {bytes_ / ms / 1e6:6.0f} GB/s", flush=True)
