#!/usr/bin/env python3
"""Timing-only ablations of the ragged-batch kernels (second formulation) on 150-base reads: which part costs what.
The ablated kernels write wrong results; only their run time is read."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
from bitnuc_amd import build as _b
ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream, lib_path=_b.ensure_built(sweep=True))  # the ablated kernels live in the evidence build
N, L = 10**9, 150
count = N // L
seq = torch.empty(N, dtype=torch.uint8, device=dev)
back = torch.empty(N + 4096, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(seq, N, 0xB17C0DE)
off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L
wo = torch.empty(count + 1, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
total = ctx.batch_word_offsets_dev(off, count, wo)
words = torch.empty(total + 64, dtype=torch.int64, device=dev)


def once(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    fn()
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b)


names = {0: "full", 1: "no window loads", 2: "no scatter/scan", 3: "no window, no scatter/scan", 4: "no partial edge chunks",
         7: "no window/scatter/edges", 8: "no record load", 9: "no record, no window", 11: "no record/window/scatter", 15: "nothing but data movement"}
res = {}
for rnd in range(7):
    for a in names:
        ctx.require_variant("batch_abl", a)
        e = once(lambda: ctx.encode_batch_dev(seq, off, wo, count, total, words)) if a in (0, 1, 2, 3, 8, 9, 11) else None
        d = once(lambda: ctx.decode_batch_dev(words, wo, off, count, total, back))
        if rnd >= 2:
            res.setdefault(a, ([], []))
            if e is not None:
                res[a][0].append(e)
            res[a][1].append(d)
ctx.require_variant("batch_abl", 0)
fe = statistics.median(once(lambda: ctx.encode_fixed_dev(seq, L, L, count, words)) for _ in range(7))
fd = statistics.median(once(lambda: ctx.decode_fixed_dev(words, L, L, count, back)) for _ in range(7))
try:
    ctx.sync()
except Exception as ex:  # ablated encodes read fake positions: a latched InvalidBase is expected noise here
    print("(sync:", ex, ")")
print("ms per launch incl. the 22 us tile-record pre-kernel; fixed-length kernels (no tables): encode %.4f decode %.4f" % (fe, fd))
for a, (e, d) in res.items():
    print(f"  abl {a:2d} {names[a]:32s} enc {statistics.median(e) if e else float('nan'):.4f}  dec {statistics.median(d):.4f}")
