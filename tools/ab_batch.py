#!/usr/bin/env python3
"""In-process A/B of the two encode_batch tile bodies (stream cut vs raw-byte funnel)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream)
N = 10**9
seq = torch.empty(N, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(seq, N, 1)
for L in (32, 100, 150, 1000):
    count = N // L
    off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L
    wo = torch.empty(count + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    total = ctx.batch_word_offsets_dev(off, count, wo)
    words = [torch.empty(total, dtype=torch.int64, device=dev) for _ in range(2)]
    res = {0: [], 1: []}
    for rnd in range(9):
        for mode in (0, 1):
            ctx.set_variant("batch_stream", mode)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
            ev[0].record(stream)
            for i in range(6):
                ctx.encode_batch_dev(seq, off, wo, count, total, words[mode])
                ev[i + 1].record(stream)
            torch.cuda.synchronize()
            res[mode].append(statistics.mean(ev[i].elapsed_time(ev[i + 1]) for i in range(2, 6)))
    ctx.sync()
    assert torch.equal(words[0], words[1])
    print(f"L={L}: raw-byte funnel {statistics.median(res[0]):.4f} ms | stream cut {statistics.median(res[1]):.4f} ms", flush=True)

for L in (100, 150, 151, 1000):
    count = N // L
    words = [torch.empty(count * ((L + 31) // 32), dtype=torch.int64, device=dev) for _ in range(2)]
    res = {0: [], 1: []}
    for rnd in range(9):
        for mode in (0, 1):
            ctx.set_variant("fixed_stream", mode)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
            ev[0].record(stream)
            for i in range(6):
                ctx.encode_fixed_dev(seq, L, L, count, words[mode])
                ev[i + 1].record(stream)
            torch.cuda.synchronize()
            res[mode].append(statistics.mean(ev[i].elapsed_time(ev[i + 1]) for i in range(2, 6)))
    ctx.sync()
    assert torch.equal(words[0], words[1])
    print(f"fixed L={L}: raw-byte funnel {statistics.median(res[0]):.4f} ms | stream cut {statistics.median(res[1]):.4f} ms", flush=True)
