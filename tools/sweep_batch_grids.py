#!/usr/bin/env python3
"""grid_mult (0 = one tile per wave / workgroup, N = N resident workgroups per CU) for the read-batch kernels, 150-base reads."""
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
L = int(sys.argv[1]) if len(sys.argv) > 1 else 150
count = N // L
seq = torch.empty(N, dtype=torch.uint8, device=dev)
back = torch.empty(N, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(seq, N, 1)
off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L
wo = torch.empty(count + 1, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
total = ctx.batch_word_offsets_dev(off, count, wo)
words = [torch.empty(total, dtype=torch.int64, device=dev) for _ in range(2)]
ctx.encode_batch_dev(seq, off, wo, count, total, words[0])
ctx.encode_batch_dev(seq, off, wo, count, total, words[1])
ctx.sync()
ops = (("encode_batch", lambda i: ctx.encode_batch_dev(seq, off, wo, count, total, words[i & 1])),
       ("decode_batch", lambda i: ctx.decode_batch_dev(words[i & 1], wo, off, count, total, back)),
       ("encode_fixed", lambda i: ctx.encode_fixed_dev(seq, L, L, count, words[i & 1])),
       ("decode_fixed", lambda i: ctx.decode_fixed_dev(words[i & 1], L, L, count, back)))
mults = (0, 4, 8, 12, 16)
rows = {}
for rnd in range(5):
    for mult in mults:
        ctx.require_variant("grid_mult", mult)
        for name, fn in ops:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
            ev[0].record(stream)
            for i in range(6):
                fn(i)
                ev[i + 1].record(stream)
            torch.cuda.synchronize()
            rows.setdefault((name, mult), []).append(statistics.mean(ev[i].elapsed_time(ev[i + 1]) for i in range(2, 6)))
ctx.sync()
for name, _ in ops:
    print(f"L={L} {name:13s} " + " | ".join(f"{m}: {statistics.median(rows[(name, m)]):.4f} ms" for m in mults), flush=True)
