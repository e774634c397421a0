#!/usr/bin/env python3
"""kmer_slide_kernel (every 31-base window of 10^9 bases -> u64): rounds per wave trip x grid form.  Sustained bursts,
interleaved rounds, one process; outputs compared with the default's."""
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
ctx.require_variant("slide_impl", 0)  # this tool is about the STRIP kernel (rounds of 992); tools/ab_r03.py compares it with the line-aligned one
N, k = 10**9, 31
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 1
seq = torch.empty(N, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(seq, N, 0xB17C0DE)
nwin = (N - k) // stride + 1
outs = [torch.empty(nwin, dtype=torch.int64, device=dev) for _ in range(2)]
ref = torch.empty(nwin, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
ctx.as_2bit_batch_dev(seq, k, stride, nwin, ref)
ctx.sync()
BURST = 4
SET = [(1, 0)] + [(u, g) for u in (2, 4, 8) for g in (0, 4, 8, 12, 16, 32)]
res = {s: [] for s in SET}
ok = {}
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
    ctx.as_2bit_batch_dev(seq, k, stride, nwin, outs[flip[0]])


for rnd in range(5):
    for s in SET:
        ctx.require_variant("slide_rounds", s[0])
        ctx.require_variant("grid_mult", s[1])
        t = once(run)
        if rnd == 0:
            outs[0].zero_()
            torch.cuda.synchronize()
            ctx.as_2bit_batch_dev(seq, k, stride, nwin, outs[0])
            ctx.sync()
            ok[s] = bool(torch.equal(outs[0], ref))
        if rnd >= 1:
            res[s].append(t)
ctx.require_variant("slide_rounds", 1)
ctx.require_variant("slide_impl", 1)
ctx.require_variant("grid_mult", 0)
alg = (N - k + 1) + 8 * nwin if stride == 1 else N + 8 * nwin
print(f"stride {stride}: {nwin} windows, {alg/1e9:.3f} GB algorithmic")
for s in SET:
    m = statistics.median(res[s])
    print(f"  {s[0]} round(s) per trip, grid {'one wave per trip' if s[1] == 0 else str(s[1]) + ' WG/CU resident'}: {m:.4f} ms  {alg/m/1e6:6.0f} GB/s  {'same output' if ok[s] else 'MISMATCH'}")
