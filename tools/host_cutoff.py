#!/usr/bin/env python3
"""Where the size dispatch of the host-pointer bulk calls belongs: bitnuc_encode / bitnuc_decode of n bases as the library's host
SWAR code (one thread) and as the GPU path (stage in, launch, stage out, one wait), same caller buffers, per-call wall time."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bitnuc_amd

ctx = bitnuc_amd.Context(0)
rng = np.random.default_rng(0)
big = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=1 << 23)]


def t(fn, reps):
    fn()
    fn()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e6


print("    bases | encode host us | encode GPU us | decode host us | decode GPU us")
for n in (1 << 14, 1 << 15, 1 << 16, 1 << 17, 1 << 18, 3 << 17, 1 << 19, 3 << 18, 1 << 20, 1 << 21, 1 << 22):
    seq = big[:n].copy()
    w = np.zeros((n + 31) // 32, dtype=np.uint64)
    back = np.zeros(n, dtype=np.uint8)
    reps = max(5, min(200, (1 << 24) // n))
    row = []
    for force in (0, 1):
        ctx.require_variant("force_gpu", force)
        ctx.require_variant("host_cutoff", 1 << 30)
        row.append((t(lambda: ctx.encode_into(seq, w), reps), t(lambda: ctx.decode_into(w, n, back), reps)))
    assert np.array_equal(back, seq)
    print(f"{n:9d} | {row[0][0]:14.1f} | {row[1][0]:13.1f} | {row[0][1]:14.1f} | {row[1][1]:13.1f}", flush=True)
