#!/usr/bin/env python3
"""Host-pointer k-mer calls (the drop-in forms of configs 3 and 5 for data in host memory): pipelined staging (host_pipe.h engine)
vs round 2's simple staged path (set_variant host_pipeline 0), caller-owned and already-touched outputs, PCIe inclusive."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bitnuc_amd
from bitnuc_amd import _lib as L

ctx = bitnuc_amd.Context(0)
lib = ctx._lib
rng = np.random.default_rng(1)


def bases(n):
    return np.repeat(np.frombuffer(b"ACGT", dtype=np.uint8)[np.frombuffer(rng.bytes(n // 4 + 1), dtype=np.uint8) & 3], 4)[:n].copy()


def timed(fn, reps=3):
    ts = []
    for _ in range(reps + 1):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return min(ts[1:])


def ptr(a):
    return C.c_void_p(a.ctypes.data)


err = L.BitnucErr()
k, cnt = 31, 10**8
km = bases(k * cnt)
out = np.zeros(cnt, dtype=np.uint64)
nwin_n = 2 * 10**8
wsrc = bases(nwin_n)
wout = np.zeros(nwin_n - k + 1, dtype=np.uint64)
n = 10**9
ref = bases(n)
dist = np.zeros(n - k + 1, dtype=np.uint8)
rows = []
for pipe in (1, 0):
    ctx.require_variant("host_pipeline", pipe)
    t1 = timed(lambda: lib.bitnuc_as_2bit_batch(ctx._h, ptr(km), k, k, cnt, ptr(out), C.byref(err)))
    chk1 = int(out[::99991].sum())
    t2 = timed(lambda: lib.bitnuc_as_2bit_batch(ctx._h, ptr(wsrc), k, 1, nwin_n - k + 1, ptr(wout), C.byref(err)))
    chk2 = int(wout[::99991].sum())
    t3 = timed(lambda: lib.bitnuc_kmer_hdist_scan(ctx._h, ptr(ref), n, k, C.c_uint64(0x0123456789ABCDEF & ((1 << 62) - 1)), ptr(dist), C.byref(err)))
    chk3 = int(dist[::9973].astype(np.int64).sum())
    rows.append((pipe, t1, t2, t3, (chk1, chk2, chk3)))
    print(f"{'pipelined' if pipe else 'simple   '}: 10^8 dense 31-mers {t1 * 1e3:7.1f} ms = {cnt / t1 / 1e9:5.2f} G k-mers/s ({(31 + 8) * cnt / t1 / 1e9:5.1f} GB/s over PCIe, both ways)"
          f"   2e8 windows -> u64 {t2 * 1e3:7.1f} ms ({9 * nwin_n / t2 / 1e9:5.1f} GB/s)   scan of 10^9 bases {t3 * 1e3:7.1f} ms = {n / t3 / 1e9:5.1f} G windows/s ({2 * n / t3 / 1e9:5.1f} GB/s)", flush=True)
assert rows[0][4] == rows[1][4], "pipelined and simple paths disagree"
print("same results on both paths")
