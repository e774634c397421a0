#!/usr/bin/env python3
"""Host-pointer forms of the packed-word analysis calls at BASELINE size: bulk hdist of two 10^9-base packed
buffers, base counts of one, per-word distances of 3.1e7 pairs / one query -- caller memory pageable, PCIe inclusive, next to the
CPU oracle's single-thread time for the same call on the same host.  Median of 5 after a warm-up call."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import bitnuc_amd
import oracle_py as oracle

ctx = bitnuc_amd.Context(0)
nb = 10**9
nw = (nb + 31) // 32
rng = np.random.default_rng(7)
a = rng.integers(0, 1 << 63, nw, dtype=np.uint64)
b = a.copy()
b[::7] ^= np.uint64(0x0123456789ABCDEF)


def med(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t)
    return statistics.median(ts), r


rows = []
t, r = med(lambda: ctx.hdist(a, b, nb))
tc, rc = med(lambda: oracle.hdist(a, b, nb), reps=3)
rows.append(("hdist(a, b, 1e9 bases)", t, tc, r == rc, 2 * nw * 8))
t, r = med(lambda: ctx.base_counts(a, nb))
tc, rc = med(lambda: oracle.base_counts(a, nb), reps=3)
rows.append(("base_counts(1e9 bases)", t, tc, r == rc, nw * 8))
t, r = med(lambda: ctx.hdist_pairs(a, b, 32))
tc, rc = med(lambda: oracle.hdist_pairs(a, b, 32), reps=3)
rows.append(("hdist_pairs(3.1e7 words)", t, tc, bool(np.array_equal(r, rc)), 2 * nw * 8 + nw))
q = int(a[12345])
t, r = med(lambda: ctx.hdist_query(q, a, 32))
tc, rc = med(lambda: oracle.hdist_pairs(a, np.full(nw, q, np.uint64), 32), reps=3)
rows.append(("hdist_query(3.1e7 words)", t, tc, bool(np.array_equal(r, rc)), nw * 8 + nw))
for name, t, tc, ok, moved in rows:
    print(f"{name:28s} library {t * 1e3:8.2f} ms ({moved / t / 1e9:5.1f} GB/s over PCIe)   CPU oracle, 1 thread {tc * 1e3:8.2f} ms   {'same result' if ok else 'MISMATCH'}")
print("host pipe:", ctx.host_pipe_info())
