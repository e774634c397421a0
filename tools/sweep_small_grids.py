#!/usr/bin/env python3
"""grid_mult (0 = one tile per workgroup, N = N resident workgroups per CU, grid-stride) for the short streaming
kernels: hdist_query, hdist_pairs, split_packed; 10^9 packed bases, cache-cold inputs."""
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
d = torch.empty(nw, dtype=torch.uint8, device=dev)
idx = n // 2 + 5
nl, nr = ctx.split_packed_sizes(nw, n, idx, canonical=True)
sl, sr = torch.empty(nl, dtype=torch.int64, device=dev), torch.empty(nr, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
ops = (("hdist_query", 9 * nw, lambda i: ctx.hdist_query_dev(0x1B1B1B1B1B1B1B1B, w[i & 1], nw, 32, d)),
       ("hdist_pairs", 17 * nw, lambda i: ctx.hdist_pairs_dev(w[i & 1], w[2 + (i & 1)], nw, 32, d)),
       ("split_packed", 8 * (nw + nl + nr), lambda i: ctx.split_packed_dev(w[i & 1], nw, n, idx, sl, sr, canonical=True)))
KEY = sys.argv[1] if len(sys.argv) > 1 else "grid_mult"
mults = tuple(int(x) for x in sys.argv[2:]) or (0, 2, 4, 8, 16, 32)
rows = {}
for rnd in range(7):
    for mult in mults:
        ctx.require_variant(KEY, mult)
        for name, _, fn in ops:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
            ev[0].record(stream)
            for i in range(8):
                fn(i)
                ev[i + 1].record(stream)
            torch.cuda.synchronize()
            rows.setdefault((name, mult), []).append(statistics.mean(ev[i].elapsed_time(ev[i + 1]) for i in range(2, 8)))
ctx.sync()
for name, nbytes, _ in ops:
    for mult in mults:
        ms = statistics.median(rows[(name, mult)])
        print(f"{name:13s} {KEY} {mult:2d}: {ms * 1e3:7.1f} us  {nbytes / ms / 1e6:6.0f} GB/s", flush=True)
