#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points (bitnuc_encode / bitnuc_decode):
never bench.py's `value`, recorded in DESIGN.md."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import bitnuc_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10**9
ctx = bitnuc_amd.Context(0)
rng = np.random.default_rng(1)
seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)]
w = ctx.encode_array(seq[: 1 << 20])  # warm up scratch
for rep in range(3):
    t0 = time.perf_counter()
    w = ctx.encode_array(seq)
    t1 = time.perf_counter()
    d = ctx.decode_array(w, n)
    t2 = time.perf_counter()
    print(f"host-pointer encode: {n/(t1-t0)/1e9:.2f} Gbases/s ({1.25*n/(t1-t0)/1e9:.1f} GB/s over PCIe+staging); "
          f"decode: {n/(t2-t1)/1e9:.2f} Gbases/s", flush=True)
assert np.array_equal(d, seq)
