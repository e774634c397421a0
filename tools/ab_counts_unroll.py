#!/usr/bin/env python3
"""base_counts_kernel: 16-byte loads in flight per thread (-DBITNUC_COUNTS_UNROLL) x resident grid (reduce_mult), 10^9 packed
bases, sustained bursts over two alternating 250 MB inputs (cache-cold).  Libraries are built next to the product's (never over it)."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bitnuc_amd
from bitnuc_amd import build as B

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
n = 10**9
libs = {u: os.path.join(ROOT, "bitnuc_amd", f"libbitnuc_hip_cu{u}.so") for u in (4, 8, 16)}
for u, p in libs.items():  # built on demand, next to the product's library and never over it
    if not os.path.exists(p):
        B.build_library(force=True, verbose=False, extra_flags=[f"-DBITNUC_COUNTS_UNROLL={u}"], out=p)
ctxs = {u: bitnuc_amd.Context(0, stream=stream.cuda_stream, lib_path=p) for u, p in libs.items()}
c0 = next(iter(ctxs.values()))
words = [torch.empty(n // 32, dtype=torch.int64, device=dev) for _ in range(2)]
seq = torch.empty(n, dtype=torch.uint8, device=dev)
for r in range(2):
    c0.nucgen_dev(seq, n, 5 + r)
    c0.encode_dev(seq, n, words[r])
c0.sync()
counts = torch.zeros(4, dtype=torch.int64, device=dev)
ref = None
res = {}
for rnd in range(6):
    for u, c in ctxs.items():
        for mult in (1, 2, 3):
            c.require_variant("reduce_mult", mult)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c.base_counts_dev(words[0], n // 32, n, counts)
            a.record(stream)
            for i in range(8):
                c.base_counts_dev(words[i & 1], n // 32, n, counts)
            b.record(stream)
            torch.cuda.synchronize()
            got = counts.tolist()
            if ref is None:
                ref = got
            assert got == ref, (u, mult, got, ref)
            if rnd:
                res.setdefault((u, mult), []).append(a.elapsed_time(b) / 8)
for (u, mult), v in sorted(res.items()):
    m = statistics.median(v)
    print(f"unroll {u:2d}, {mult} workgroups per CU: {m * 1e3:6.1f} us  {0.25 * n / m / 1e6:6.0f} GB/s  {0.25 * n / m / 1e6 / 8000:.3f}")
