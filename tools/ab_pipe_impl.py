#!/usr/bin/env python3
"""Host-pointer path: the staged engine (own pinned buffers, 8 + 4 copy threads) against the direct engine (pageable copies from the
calling thread and one mover thread) -- csrc/host_pipe.h.  10^9 bases, caller-owned pageable arrays, interleaved rounds in one
process, median of 5 per cell; every result compared between the engines (and the round trip with the input)."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bitnuc_amd

ctx = bitnuc_amd.Context(0)
n = 10**9
nw = (n + 31) // 32
rng = np.random.default_rng(3)
seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)]
words = [np.zeros(nw, dtype=np.uint64) for _ in range(2)]
back = [np.zeros(n, dtype=np.uint8) for _ in range(2)]
k = 31
nk = n // k  # dense 31-mers of the same 10^9 bases
kout = [np.zeros(nk, dtype=np.uint64) for _ in range(2)]
dist = [np.zeros(n - k + 1, dtype=np.uint8) for _ in range(2)]
lib, C, L = ctx._lib, __import__("ctypes"), bitnuc_amd._lib


def call(fn, *a):
    err = L.BitnucErr()
    st = fn(ctx._h, *a, C.byref(err))
    assert st == 0, st


P = lambda a: C.c_void_p(a.ctypes.data)
CASES = {
    "encode 1e9 bases": lambda i: ctx.encode_into(seq, words[i]),
    "decode 1e9 bases": lambda i: ctx.decode_into(words[0], n, back[i]),
    "as_2bit_batch 3.2e7 dense 31-mers": lambda i: call(lib.bitnuc_as_2bit_batch, P(seq), k, k, nk, P(kout[i])),
    "kmer_hdist_scan 1e9 bases": lambda i: call(lib.bitnuc_kmer_hdist_scan, P(seq), n, k, C.c_uint64(0x1B1B1B1B1B1B1B), P(dist[i])),
}
UNITS = {"encode 1e9 bases": n, "decode 1e9 bases": n, "as_2bit_batch 3.2e7 dense 31-mers": nk, "kmer_hdist_scan 1e9 bases": n - k + 1}
res = {}
ctx.require_variant("pipe_impl", 0)
ctx.encode_into(seq, words[0])
for name, fn in CASES.items():
    for impl in (0, 1):  # warm-up + first touch of every output array
        ctx.require_variant("pipe_impl", impl)
        fn(impl)
    for rnd in range(5):
        for impl in (0, 1):
            ctx.require_variant("pipe_impl", impl)
            t = time.perf_counter()
            fn(impl)
            res.setdefault((name, impl), []).append(time.perf_counter() - t)
same = {"encode 1e9 bases": np.array_equal(words[0], words[1]), "decode 1e9 bases": np.array_equal(back[0], back[1]) and np.array_equal(back[1], seq),
        "as_2bit_batch 3.2e7 dense 31-mers": np.array_equal(kout[0], kout[1]), "kmer_hdist_scan 1e9 bases": np.array_equal(dist[0], dist[1])}
for name in CASES:
    a, b = statistics.median(res[(name, 0)]), statistics.median(res[(name, 1)])
    u = UNITS[name]
    print(f"{name:34s} staged {a * 1e3:7.2f} ms ({u / a / 1e9:5.1f} G/s; {min(res[(name, 0)]) * 1e3:7.2f}..{max(res[(name, 0)]) * 1e3:7.2f})   "
          f"direct {b * 1e3:7.2f} ms ({u / b / 1e9:5.1f} G/s; {min(res[(name, 1)]) * 1e3:7.2f}..{max(res[(name, 1)]) * 1e3:7.2f})   {'same result' if same[name] else 'MISMATCH'}")
print(ctx.host_pipe_info())
