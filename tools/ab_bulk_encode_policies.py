#!/usr/bin/env python3
"""Bulk encode variants in ENCODE-ONLY sustained bursts (two output buffers): what the store policy and the workgroup size do
to the kernel by itself, as opposed to inside the encode+decode pair of the timed step (tools/sweep_pairs.py).
Variant = (UNROLL, BLOCK, nt loads, nt stores, LDS transpose, XCD order): see BITNUC_VARIANTS in csrc/codec.hip."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd
from bitnuc_amd import build as _build

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream, lib_path=_build.ensure_built(sweep=True))
N = 10**9
seq = torch.empty(N, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(seq, N, 0xB17C0DE)
outs = [torch.empty(N // 32 + 64, dtype=torch.int64, device=dev) for _ in range(2)]
VARIANTS = {14: "U2 B128 nt-load plain-store (shipped)", 43: "U2 B128 nt-load plain-store, XCD order", 3: "U2 B256 nt-load plain-store", 2: "U2 B256 nt-load nt-store",
            39: "U2 B128 nt-load nt-store, XCD order", 35: "U4 B128 nt-load nt-store", 37: "U4 B128 nt-load plain-store", 4: "U2 B256 plain-load plain-store",
            22: "U2 B256 plain-load nt-store", 36: "U2 B64 nt-load plain-store", 38: "U1 B128 nt-load plain-store"}
BURST = 12
flip = [0]


def once(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    a.record(stream)
    for _ in range(BURST):
        fn()
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / BURST


def run():
    flip[0] ^= 1
    ctx.encode_dev(seq, N, outs[flip[0]])


res = {v: [] for v in VARIANTS}
for rnd in range(6):
    for v in VARIANTS:
        ctx.require_variant("encode", v)
        t = once(run)
        if rnd >= 1:
            res[v].append(t)
ctx.require_variant("encode", 14)
print("bulk encode of 10^9 bases in encode-only sustained bursts (1.25 GB algorithmic per launch)")
for v in sorted(VARIANTS, key=lambda v: statistics.median(res[v])):
    m = statistics.median(res[v])
    print(f"  e{v:<3d} {VARIANTS[v]:44s} {m:.4f} ms  {1.25e9/m/1e6:6.0f} GB/s")
