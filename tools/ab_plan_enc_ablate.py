#!/usr/bin/env python3
"""Why the plan encode's DENSE tiles (32-base reads: every tile is a plain 2 KiB bulk tile) run below the bulk encode kernel:
timing-only ablations of its extra loads (evidence build).  For 32-base reads every ablation still yields the right words.
Sustained bursts, two output buffers, interleaved rounds, one process."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd
from bitnuc_amd import build as _build

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream, lib_path=_build.ensure_built(sweep=True))
N, L = 10**9, 32
count = N // L
seq = torch.empty(N, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(seq, N, 0xB17C0DE)
off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L
torch.cuda.synchronize()
plan = bitnuc_amd.BatchPlan(ctx, off, count)
total = plan.total_words
outs = [torch.empty(total + 64, dtype=torch.int64, device=dev) for _ in range(2)]
bws = [torch.empty(N // 32 + 64, dtype=torch.int64, device=dev) for _ in range(2)]  # alternate: one 250 MB output would live in the Infinity Cache
ref = torch.empty(total + 64, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
plan.encode_dev(seq, ref)
ctx.sync()
BLOCKS = (256, 128, 64)
NAMES = {0: "as shipped", 1: "tile base by arithmetic (no tile_base load)", 2: "no pad-byte load", 4: "no 129th-chunk load", 6: "no pad-byte and no 129th-chunk load",
         7: "none of the three (two chunk loads per lane only)"}
BURST = 12
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
    plan.encode_dev(seq, outs[flip[0]])


res = {k: [] for k in NAMES}
res["bulk"] = []
for b in BLOCKS[1:]:
    res[("block", b)] = []
ok = {}
for rnd in range(7):
    for abl in NAMES:
        ctx.require_variant("plan_enc_abl", abl)
        t = once(run)
        if rnd == 0:
            outs[0].zero_()
            torch.cuda.synchronize()
            plan.encode_dev(seq, outs[0])
            ctx.sync()
            ok[abl] = bool(torch.equal(outs[0][:total], ref[:total]))
        if rnd >= 2:
            res[abl].append(t)
    ctx.require_variant("plan_enc_abl", 0)
    for b in BLOCKS[1:]:
        ctx.require_variant("plan_enc_block", b)
        t = once(run)
        if rnd >= 2:
            res[("block", b)].append(t)
    ctx.require_variant("plan_enc_block", 256)
    u = once(lambda: ctx.encode_dev(seq, N, bws[flip.__setitem__(0, flip[0] ^ 1) or flip[0]]))
    if rnd >= 2:
        res["bulk"].append(u)
alg = N + 8 * total
print(f"plan encode of {count} reads of 32 bases (every tile dense and aligned), {alg/1e9:.3f} GB algorithmic")
for abl in NAMES:
    m = statistics.median(res[abl])
    print(f"  {NAMES[abl]:52s} {m:.4f} ms  {alg/m/1e6:6.0f} GB/s  {'same words' if ok[abl] else 'MISMATCH'}")
for b in BLOCKS[1:]:
    m = statistics.median(res[("block", b)])
    print(f"  {'as shipped, ' + str(b) + '-thread workgroups':52s} {m:.4f} ms  {alg/m/1e6:6.0f} GB/s")
m = statistics.median(res["bulk"])
print(f"  {'bulk encode_kernel on the same bytes':52s} {m:.4f} ms  {alg/m/1e6:6.0f} GB/s")
