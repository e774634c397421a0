#!/usr/bin/env python3
"""What does this memory system give the every-window kernel's traffic (1 B read : 8 B written)?  No-arithmetic kernels of
that shape (evidence build, stream probe mode 5) in sustained bursts with two alternating 8 GB outputs, next to the pure
fill / copy probes and the real kernel.  Variants: nt vs plain stores, 1 / 2 / 4 rounds per trip, the wave's eight 1 KiB stores
consecutive (what kmer_slide2_kernel does) or interleaved KiB by KiB with the other waves of the workgroup."""
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
outs = [torch.empty(N - 30, dtype=torch.int64, device=dev) for _ in range(2)]
ctx.sync()
BURST, ROUNDS = 4, 5


def sustained(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(0)
    a.record(stream)
    for i in range(BURST):
        fn(i)
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / BURST


cases = {}
nbytes = (N >> 12) << 12
for nts in (1, 0):
    for mp in (0, 1):
        for ulog in (0, 1, 2):
            mode = 5 | (16 if nts else 0) | (32 if mp else 0) | (ulog << 6)
            cases[f"shape probe: {'nt' if nts else 'plain'} stores, {'interleaved KiB' if mp else '8 consecutive KiB per wave'}, {1 << ulog} round(s)/trip"] = \
                (lambda i, mode=mode: ctx.stream_probe_dev(mode, seq, outs[i & 1], nbytes), 9 * nbytes)
cases["fill probe (nt), 8 GB"] = (lambda i: ctx.stream_probe_dev(2 | 16, None, outs[i & 1], 8 * nbytes), 8 * nbytes)
cases["fill probe (plain), 8 GB"] = (lambda i: ctx.stream_probe_dev(2, None, outs[i & 1], 8 * nbytes), 8 * nbytes)
for pol in (3, 1):
    for u in (1, 2, 4):
        def real(i, u=u, pol=pol):
            ctx.require_variant("slide2_rounds", u)
            ctx.require_variant("dense_policy", pol)
            ctx.as_2bit_batch_dev(seq, 31, 1, N - 30, outs[i & 1])
        cases[f"kmer_slide2_kernel, {'nt' if pol & 2 else 'plain'} stores, {u} round(s)/trip"] = (real, N + 8 * (N - 30))
res = {k: [] for k in cases}
for rnd in range(ROUNDS + 1):
    for k, (fn, _) in cases.items():
        t = sustained(fn)
        if rnd:
            res[k].append(t)
for k, v in res.items():
    m = statistics.median(v)
    alg = cases[k][1]
    print(f"{k:78s} {m:.4f} ms  {alg / m / 1e6:6.0f} GB/s  {alg / m / 1e6 / 8000:.3f}")
