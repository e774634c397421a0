#!/usr/bin/env python3
"""Small-call latency of the host-pointer API (BASELINE configs[0]: a 1000-base sequence)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bitnuc_amd

ctx = bitnuc_amd.Context(0)
rng = np.random.default_rng(0)
s1000 = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=1000)]
w = ctx.encode_array(s1000)


def t(fn, n=300):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


print(f"as_2bit(31-mer)            {t(lambda: ctx.as_2bit(s1000[:31])):8.1f} us/call")
print(f"encode(1000 bases)         {t(lambda: ctx.encode_array(s1000)):8.1f} us/call")
print(f"decode(1000 bases)         {t(lambda: ctx.decode_array(w, 1000)):8.1f} us/call")
reads = [bytes(s1000[:150])] * 10000
print(f"encode_many(10000 x 150)   {t(lambda: ctx.encode_many(reads), 5):8.1f} us/call ({t(lambda: ctx.encode_many(reads), 5)/10000:.3f} us/read)")
