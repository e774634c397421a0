#!/usr/bin/env python3
"""Plan encode by workgroup size (a wave owns a tile, so any number of waves per workgroup works) on L-base reads.
Sustained bursts, two output buffers, interleaved rounds, one process.  usage: ab_plan_block.py [L ...]"""
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
N = 10**9
seq = torch.empty(N, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(seq, N, 0xB17C0DE)
BURST = 12
BLOCKS = (256, 128, 64)


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
    torch.cuda.synchronize()
    flip = [0]

    def run():
        flip[0] ^= 1
        plan.encode_dev(seq, outs[flip[0]])

    res = {b: [] for b in BLOCKS}
    for rnd in range(8):
        for b in BLOCKS:
            ctx.require_variant("plan_enc_block", b)
            t = once(run)
            if rnd >= 2:
                res[b].append(t)
    ctx.require_variant("plan_enc_block", 256)
    alg = L * count + 8 * total
    print(f"L={L}: " + " | ".join(f"{b} threads {statistics.median(res[b]):.4f} ms {alg/statistics.median(res[b])/1e6:6.0f} GB/s" for b in BLOCKS), flush=True)
    plan.close()
    del plan, outs, off
