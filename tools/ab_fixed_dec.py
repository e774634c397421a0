#!/usr/bin/env python3
"""In-process A/B of decode_fixed on back-to-back reads: byte scatter into LDS (fixed_dec_strip=0), bit strip with per-lane
64-bit positions (1), the shared tile body of the plan decode with 32-bit tile-relative positions and the dense fast
path (2).  Input words alternate between two copies (cache-cold)."""
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
N = 10**9
seq = torch.empty(N, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(seq, N, 1)
for L in [int(x) for x in sys.argv[1:]] or [16, 31, 32, 36, 100, 150, 151, 160, 250, 1000, 100000]:
    count = N // L
    wpr = (L + 31) // 32
    words = torch.empty(count * wpr, dtype=torch.int64, device=dev)
    ctx.encode_fixed_dev(seq, L, L, count, words)
    words2 = words.clone()
    outs = [torch.zeros(count * L, dtype=torch.uint8, device=dev) for _ in range(3)]
    torch.cuda.synchronize()
    res = {0: [], 1: [], 2: []}
    for rnd in range(9):
        for mode in (0, 1, 2):
            ctx.require_variant("fixed_dec_strip", mode)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
            ev[0].record(stream)
            for i in range(6):
                ctx.decode_fixed_dev(words if i & 1 else words2, L, L, count, outs[mode])
                ev[i + 1].record(stream)
            torch.cuda.synchronize()
            res[mode].append(statistics.mean(ev[i].elapsed_time(ev[i + 1]) for i in range(2, 6)))
    ctx.sync()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and torch.equal(outs[0], seq[: count * L])
    ctx.require_variant("fixed_dec_strip", 2)
    alg = count * L + 8 * count * wpr
    m0, m1, m2 = statistics.median(res[0]), statistics.median(res[1]), statistics.median(res[2])
    print(f"L={L}: byte scatter {m0:.4f} ms ({alg / m0 / 1e6:.0f} GB/s) | bit strip {m1:.4f} ms ({alg / m1 / 1e6:.0f} GB/s) | shared tile body {m2:.4f} ms ({alg / m2 / 1e6:.0f} GB/s)", flush=True)
