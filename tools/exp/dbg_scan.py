import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "./tests")
import numpy as np
import bitnuc_amd, oracle_py
ctx = bitnuc_amd.Context(0); ctx.set_variant("force_gpu", 1)
rng = np.random.default_rng(1)
for unroll in (1, 2, 4):
    ctx.set_variant("scan_unroll", unroll)
    for k in (2, 31):
        for n in (2080, 4128, 5000):
            s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)]
            q = int(rng.integers(0, 1 << 62))
            got = ctx.kmer_hdist_scan(s, k, q); exp = oracle_py.kmer_hdist_scan(s, k, q)
            bad = np.nonzero(got != exp)[0]
            print(unroll, k, n, "mismatches:", bad[:20], len(bad))
