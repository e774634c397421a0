#!/usr/bin/env python3
"""encode_batch_plan_kernel: consecutive tiles per wave trip (1, 2, 4) on L-base reads, next to the bulk encode kernel.
Sustained bursts alternating two output buffers, interleaved rounds, one process; every setting's words are compared.
usage: ab_plan_enc_tiles.py [L ...]"""
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
ctx.nucgen_dev(seq, N, 0xB17C0DE)
bws = [torch.empty(N // 32 + 64, dtype=torch.int64, device=dev) for _ in range(2)]  # alternate: one 250 MB output would live in the Infinity Cache
BURST = 12
SETTINGS = [1, 2, 4]


def once(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    a.record(stream)
    for _ in range(BURST):
        fn()
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / BURST


for L in [int(a) for a in sys.argv[1:]] or [150]:
    count = N // L
    off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L
    torch.cuda.synchronize()
    plan = bitnuc_amd.BatchPlan(ctx, off, count)
    total = plan.total_words
    outs = [torch.empty(total + 64, dtype=torch.int64, device=dev) for _ in range(2)]
    ref = torch.empty(total + 64, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.require_variant("plan_enc_tiles", 1)
    plan.encode_dev(seq, ref)
    ctx.sync()
    flip = [0]

    def alt():
        flip[0] ^= 1
        return outs[flip[0]]

    res = {s: [] for s in SETTINGS}
    res["bulk"] = []
    ok = {}
    for rnd in range(7):
        for s in SETTINGS:
            ctx.require_variant("plan_enc_tiles", s)
            t = once(lambda: plan.encode_dev(seq, alt()))
            if rnd == 0:
                outs[0].zero_()
                torch.cuda.synchronize()
                plan.encode_dev(seq, outs[0])
                ctx.sync()
                ok[s] = bool(torch.equal(outs[0][:total], ref[:total]))
            if rnd >= 2:
                res[s].append(t)
        u = once(lambda: ctx.encode_dev(seq, N, bws[flip.__setitem__(0, flip[0] ^ 1) or flip[0]]))
        if rnd >= 2:
            res["bulk"].append(u)
    ctx.require_variant("plan_enc_tiles", 1)
    alg = L * count + 8 * total
    print(f"L={L}: plan encode of {count} reads, {alg/1e9:.4f} GB algorithmic")
    for s in SETTINGS:
        m = statistics.median(res[s])
        print(f"  {s} tile(s) per wave trip               {m:.4f} ms  {alg/m/1e6:6.0f} GB/s  {'same words' if ok[s] else 'MISMATCH'}")
    m = statistics.median(res["bulk"])
    print(f"  bulk encode_kernel (10^9 bases)      {m:.4f} ms  {1.25e9/m/1e6:6.0f} GB/s", flush=True)
    plan.close()
    del plan, outs, ref, off
