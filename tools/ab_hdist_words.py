#!/usr/bin/env python3
"""hdist_query / hdist_pairs over 3.1e7 packed words: four contiguous words per lane vs coalesced loads + bpermute, sustained bursts
over two alternating inputs (cache-cold)."""
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
words = [torch.empty(nw, dtype=torch.int64, device=dev) for _ in range(3)]
for r in range(3):
    ctx.nucgen_dev(seq, n, 11 + r)
    ctx.encode_dev(seq, n, words[r])
ctx.sync()
outs = [torch.empty(nw, dtype=torch.uint8, device=dev) for _ in range(2)]
res = {}
refs = {}
for rnd in range(6):
    for impl in (0, 1):
        ctx.require_variant("hdist_words_impl", impl)
        for name, fn, alg in (("query", lambda i: ctx.hdist_query_dev(0x1234567890ABCDEF, words[i & 1], nw, 32, outs[i & 1]), 9 * nw),
                              ("pairs", lambda i: ctx.hdist_pairs_dev(words[i & 1], words[2], nw, 32, outs[i & 1]), 17 * nw)):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn(0)
            a.record(stream)
            for i in range(8):
                fn(i)
            b.record(stream)
            torch.cuda.synchronize()
            chk = int(outs[1].to(torch.int64).sum().item())
            assert refs.setdefault(name, chk) == chk, (name, impl)
            if rnd:
                res.setdefault((name, impl, alg), []).append(a.elapsed_time(b) / 8)
for (name, impl, alg), v in sorted(res.items()):
    m = statistics.median(v)
    print(f"{name}: {'coalesced loads + bpermute' if impl else '4 contiguous words per lane  '}: {m * 1e3:6.1f} us  {alg / m / 1e6:6.0f} GB/s  {alg / m / 1e6 / 8000:.3f}")
