#!/usr/bin/env python3
"""Cache-policy / grid sweep for the k-mer batch (config 3) and scan (config 5) kernels,
sustained bursts of the same kernel."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
from bitnuc_amd import build as _build

ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream, lib_path=_build.ensure_built(sweep=True))  # the evidence build holds the alternatives
n, k, count = 10**9, 31, 10**8
ref = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
dist = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
kseq = torch.empty(count * k, dtype=torch.uint8, device=dev)
kout = [torch.empty(count, dtype=torch.int64, device=dev) for _ in range(2)]
for i, r in enumerate(ref):
    ctx.nucgen_dev(r, n, 0xB17C0DE + i)
ctx.nucgen_dev(kseq, count * k, 77)
ctx.sync()


def burst(fn, reps=8):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record(stream)
    for i in range(reps):
        fn(i)
        ev[i + 1].record(stream)
    torch.cuda.synchronize()
    return statistics.mean(ev[i].elapsed_time(ev[i + 1]) for i in range(2, reps))


rows = []
for rnd in range(5):
    for g in (64, 128, 256):  # threads per workgroup (grid: one item per wave)
        ctx.require_variant("grid_mult", 0)
        ctx.require_variant("kmer_block", g)
        for pol in (3,):
            ctx.require_variant("scan_policy", pol)
            for un in (2, 4):
                ctx.require_variant("scan_unroll", un)
                ms = burst(lambda i: ctx.kmer_hdist_scan_dev(ref[i % 2], n, k, 0x1B1B1B1B1B1B1B1B & ((1 << 62) - 1), dist[i % 2]))
                rows.append(("scan", g, pol * 10 + un, ms))
            ctx.require_variant("dense_policy", pol)
            for un in (1, 2):
                ctx.require_variant("dense_unroll", un)
                ms = burst(lambda i: ctx.as_2bit_batch_dev(kseq, k, k, count, kout[i % 2]))
                rows.append(("dense", g, pol * 10 + un, ms))
ctx.sync()
agg = {}
for kind, g, pol, ms in rows:
    agg.setdefault((kind, g, pol), []).append(ms)
for kind, nbytes in (("scan", 2 * (n - k + 1)), ("dense", count * (k + 8))):
    print(f"== {kind}: ms  GB/s  threads/workgroup policy*10+unroll")
    for (kd, g, pol), v in sorted(((kk, vv) for kk, vv in agg.items() if kk[0] == kind), key=lambda kv: statistics.median(kv[1])):
        ms = statistics.median(v)
        print(f"{ms:.4f} {nbytes/ms/1e6:7.0f}  g{g:<2d} p{pol}")
