#!/usr/bin/env python3
"""Bulk hdist of two 10^9-base packed buffers: grid-stride at thread granularity (4 loads a whole grid apart) vs at tile granularity
(16 KiB of each operand per workgroup trip, the read probe's walk), resident grids of 2..16 workgroups per CU; sustained bursts."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd
from bitnuc_amd import build

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream, lib_path=build.ensure_built(sweep=True))  # the selectors below exist in the evidence build only
n = 10**9
nw = n // 32
seq = torch.empty(n, dtype=torch.uint8, device=dev)
words = [torch.empty(nw, dtype=torch.int64, device=dev) for _ in range(4)]
for r in range(4):
    ctx.nucgen_dev(seq, n, 31 + r)
    ctx.encode_dev(seq, n, words[r])
ctx.sync()
res1 = torch.zeros(1, dtype=torch.int32, device=dev)
ref = None
res = {}
for rnd in range(5):
    for tiled in (0, 1):
        for mult in (1, 2, 4, 8, 16):
            ctx.require_variant("hdist_tiled", tiled)
            ctx.require_variant("hdist_mult", mult)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ctx.hdist_dev(words[0], nw, words[1], nw, n, res1)
            a.record(stream)
            for i in range(8):
                ctx.hdist_dev(words[(i & 1) * 2], nw, words[(i & 1) * 2 + 1], nw, n, res1)
            b.record(stream)
            torch.cuda.synchronize()
            got = int(res1.item())
            ref = got if ref is None else ref
            assert got == ref
            if rnd:
                res.setdefault((tiled, mult), []).append(a.elapsed_time(b) / 8)
for (tiled, mult), v in sorted(res.items()):
    m = statistics.median(v)
    print(f"{'tile walk  ' if tiled else 'thread walk'}, {mult:2d} workgroups per CU: {m * 1e3:6.1f} us  {16 * nw / m / 1e6:6.0f} GB/s  {16 * nw / m / 1e6 / 8000:.3f}")
