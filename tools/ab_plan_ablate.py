#!/usr/bin/env python3
"""decode_batch_plan_kernel knobs (tiles per wave trip, whole-chunk store policy) on L-base reads, next to the fixed-length and
bulk decode kernels.  Sustained bursts, interleaved rounds, one process."""
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
N, L = 10**9, int(sys.argv[1]) if len(sys.argv) > 1 else 150
count = N // L
seq = torch.empty(N, dtype=torch.uint8, device=dev)
back = torch.empty(N + 4096, dtype=torch.uint8, device=dev)
back2 = torch.empty(N + 4096, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(seq, N, 0xB17C0DE)
off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L
torch.cuda.synchronize()
plan = bitnuc_amd.BatchPlan(ctx, off, count)
total = plan.total_words
words = torch.empty(total + 64, dtype=torch.int64, device=dev)
bw = torch.empty(N // 32 + 64, dtype=torch.int64, device=dev)
bw2 = torch.empty(N // 32 + 64, dtype=torch.int64, device=dev)  # inputs alternate too: 250 MB would live in the Infinity Cache
plan.encode_dev(seq, words)
ctx.encode_dev(seq, N, bw)
ctx.encode_dev(seq, N, bw2)
ctx.sync()


BURST = 12


def once(fn):
    """ms per launch over a burst of back-to-back launches (sustained rate: no host sync between launches)."""
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    a.record(stream)
    for _ in range(BURST):
        fn()
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / BURST


names = {}
edges = {(u, pol): f"{u} tile(s) per trip, " + {0: "all chunks nt", 1: "all chunks plain", 2: "edge lines plain, rest nt"}[pol] for u in (1, 2, 4) for pol in (2, 0)}
res = {k: [] for k in ["fixed", "bulk"] + [("e", e) for e in edges]}
flip = [0]


def alt():
    flip[0] ^= 1
    return back if flip[0] else back2


for rnd in range(7):
    for e in edges:
        ctx.require_variant("plan_tiles", e[0])
        ctx.require_variant("plan_store", e[1])
        t = once(lambda: plan.decode_dev(words, alt()))
        if rnd >= 2:
            res[("e", e)].append(t)
    ctx.require_variant("plan_store", 2)
    ctx.require_variant("plan_tiles", 1)
    t = once(lambda: ctx.decode_fixed_dev(words, L, L, count, alt()))
    u = once(lambda: ctx.decode_dev(bw if flip[0] else bw2, N // 32, N, alt()))
    if rnd >= 2:
        res["fixed"].append(t)
        res["bulk"].append(u)
alg = L * count + 8 * total
print(f"L={L}: decode of {count} reads, {alg/1e9:.4f} GB algorithmic; bulk decode of 10^9 bases: 1.25 GB")
for e in edges:
    m = statistics.median(res[("e", e)])
    print(f"  plan decode: {edges[e]:48s} {m:.4f} ms  {alg/m/1e6:6.0f} GB/s")
plan.decode_dev(words, back)
ctx.sync()
print("  round trip:", "ok" if bool(torch.equal(back[:L * count], seq[:L * count])) else "MISMATCH")
m = statistics.median(res["fixed"])
print(f"  decode_fixed                        {m:.4f} ms  {alg/m/1e6:6.0f} GB/s")
m = statistics.median(res["bulk"])
print(f"  bulk decode_kernel (10^9 bases)     {m:.4f} ms  {1.25e9/m/1e6:6.0f} GB/s")
