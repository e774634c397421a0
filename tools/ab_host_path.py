#!/usr/bin/env python3
"""Pipelined host-pointer encode / decode of 10^9 bases: this build vs another build of the library (normally the previous commit's,
bitnuc_amd/libbitnuc_hip_prev.so), interleaved in one process, next to the box's pinned hipMemcpyAsync rates."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bitnuc_amd

n = 10**9
pin = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
devb = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
rates = {}
for name, (dst, src) in (("h2d", (devb, pin)), ("d2h", (pin, devb))):
    ts = []
    for _ in range(4):
        torch.cuda.synchronize()
        t = time.perf_counter()
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    rates[name] = (1 << 30) / min(ts[1:]) / 1e9
del pin, devb
seq = np.repeat(np.frombuffer(b"ACGT", dtype=np.uint8)[np.frombuffer(np.random.default_rng(1).bytes(n // 4 + 1), dtype=np.uint8) & 3], 4)[:n].copy()
ctxs = {"this": bitnuc_amd.Context(0), "prev": bitnuc_amd.Context(0, lib_path=os.path.join(ROOT, "bitnuc_amd", "libbitnuc_hip_prev.so"))}
w = np.zeros((n + 31) // 32, dtype=np.uint64)
back = np.zeros(n, dtype=np.uint8)
res = {(k, op): [] for k in ctxs for op in ("enc", "dec")}
for rnd in range(6):
    for k, c in ctxs.items():
        t = time.perf_counter()
        c.encode_into(seq, w)
        e = time.perf_counter() - t
        t = time.perf_counter()
        c.decode_into(w, n, back)
        d = time.perf_counter() - t
        if rnd >= 1:
            res[(k, "enc")].append(e)
            res[(k, "dec")].append(d)
assert np.array_equal(back, seq)
print(f"pinned hipMemcpyAsync: H2D {rates['h2d']:.1f} GB/s, D2H {rates['d2h']:.1f} GB/s; cores visible {len(os.sched_getaffinity(0))}")
for k in ctxs:
    e, d = statistics.median(res[(k, "enc")]), statistics.median(res[(k, "dec")])
    print(f"  {k}: encode {n/e/1e9:5.1f} Gbases/s = {100*n/e/1e9/rates['h2d']:.1f} % of pinned H2D | decode {n/d/1e9:5.1f} Gbases/s = {100*n/d/1e9/rates['d2h']:.1f} % of pinned D2H")
