#!/usr/bin/env python3
"""In-process A/B of the scan kernels: rounds of 992 windows (kmer_scan_kernel) vs line-aligned rounds of 1024
(kmer_scan2_kernel), cache policies and rounds in flight, and the fused d <= tau count.  Sustained bursts, 10^9 bases, k = 31."""
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
n, k = 10**9, 31
q = 0x1B1B1B1B1B1B1B1B & ((1 << 62) - 1)
refs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
dists = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
for i, r in enumerate(refs):
    ctx.nucgen_dev(r, n, 0xB17C0DE + i)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
ctx.sync()
BURST = 8
flip = [0]


def burst(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    a.record(stream)
    for _ in range(BURST):
        fn()
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / BURST


def scan():
    flip[0] ^= 1
    ctx.kmer_hdist_scan_dev(refs[flip[0]], n, k, q, dists[flip[0]])


def count():
    flip[0] ^= 1
    ctx.kmer_hdist_count_dev(refs[flip[0]], n, k, q, 8, cnt)


configs = [(impl, pol, un) for impl in (0, 1) for pol in (3, 1) for un in (2, 4)]
res = {c: [] for c in configs}
res["count"] = []
for rnd in range(6):
    for c in configs:
        ctx.require_variant("scan_impl", c[0])
        ctx.require_variant("scan_policy", c[1])
        ctx.require_variant("scan_unroll", c[2])
        t = burst(scan)
        if rnd:
            res[c].append(t)
    t = burst(count)
    if rnd:
        res["count"].append(t)
nwin = n - k + 1
for c in configs:
    m = statistics.median(res[c])
    print(f"impl {c[0]} ({'1024-window aligned rounds' if c[0] else '992-window rounds          '}) policy {c[1]} unroll {c[2]}: {m:.4f} ms  {2*nwin/m/1e6:6.0f} GB/s  {nwin/m/1e6:5.0f} Gwin/s")
m = statistics.median(res["count"])
print(f"fused d<=tau count: {m:.4f} ms  {nwin/m/1e6:6.0f} GB/s (1 B/window)  {nwin/m/1e6:5.0f} Gwin/s   count={int(cnt.item())}")
