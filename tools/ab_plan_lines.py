#!/usr/bin/env python3
"""Plan decode: line-owning tiles (decode_batch_plan_lines_kernel, plan_dec_lines = 1) against word tiles with shared edge lines
(decode_batch_plan_kernel, 0) on batches of L-base reads and on ragged mixes; evidence build, one process, interleaved rounds,
sustained bursts, two output buffers.  Every pair of outputs is compared (and with the input: decode(encode(x)) == x).
usage: ab_plan_lines.py [L ...]   (default 150 100 151 64 1000 250 36 and two ragged mixes)"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd
from bitnuc_amd import build

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream, lib_path=build.ensure_built(sweep=True))
N = 10**9
seq = torch.empty(N + 64, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(seq, N + 64, 0xB17C0DE)
BURST = 12


def once(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    a.record(stream)
    for _ in range(BURST):
        fn()
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / BURST


def case(name, off, count, shift=0):
    """off: int64 offsets tensor (count + 1); shift: the batch starts `shift` bytes into the buffers (alignment of the output)."""
    torch.cuda.synchronize()
    plan = bitnuc_amd.BatchPlan(ctx, off, count)
    total = plan.total_words
    nbytes = int(off[-1].item())
    words = torch.empty(total, dtype=torch.int64, device=dev)
    plan.encode_dev(seq, words)
    outs = [torch.zeros(nbytes + 256, dtype=torch.uint8, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    flip = [0]

    def run():
        flip[0] ^= 1
        plan.decode_dev(words, outs[flip[0]])
    res = {0: [], 1: [], 2: []}
    same = True
    for rnd in range(8):
        for impl in (0, 1, 2):
            ctx.require_variant("plan_dec_lines", impl)
            t = once(run)
            if rnd >= 2:
                res[impl].append(t)
        if rnd == 0:  # both buffers now hold impl 1's result in [.. flip ..]; compare each form with the input once
            for impl in (0, 1, 2):
                ctx.require_variant("plan_dec_lines", impl)
                outs[0].fill_(0xEE)
                plan.decode_dev(words, outs[0])
                ctx.sync()
                first = int(off[0].item())
                ok = torch.equal(outs[0][first:nbytes], seq[first:nbytes]) and bool((outs[0][:first] == 0xEE).all()) and bool((outs[0][nbytes:] == 0xEE).all())
                same = same and ok
    ctx.require_variant("plan_dec_lines", 0)
    alg = (nbytes - int(off[0].item())) + 8 * total
    m0, m1, m2 = statistics.median(res[0]), statistics.median(res[1]), statistics.median(res[2])
    print(f"{name:28s} {alg/1e9:.4f} GB  word tiles {m0:.4f} ms {alg/m0/1e6:6.0f} GB/s | line-owning {m1:.4f} ms {alg/m1/1e6:6.0f} GB/s {100*(m0/m1-1):+5.1f} % | chunk-owning {m2:.4f} ms {alg/m2/1e6:6.0f} GB/s {100*(m0/m2-1):+5.1f} %  "
          f"{'outputs == input, nothing outside' if same else 'MISMATCH'}", flush=True)
    plan.close()
    return same


ok = True
args = [int(a) for a in sys.argv[1:]] or [150, 100, 151, 64, 1000, 250, 36]
for L in args:
    count = N // L
    off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L
    ok = case(f"L={L}", off, count) and ok
    if L == 150:  # the same batch 1, 16, 77 bytes into the buffer: lines and chunks fall elsewhere
        for sh in (1, 16, 77):
            ok = case(f"L=150, first base at byte {sh}", off + sh, count) and ok
if not sys.argv[1:]:
    g = torch.Generator(device="cpu").manual_seed(5)
    for name, lo, hi in (("ragged 1..300", 1, 301), ("ragged 100..260", 100, 261), ("tiny 1..8 (coverage fails)", 1, 9), ("with empties 0..40", 0, 41)):
        lens = torch.randint(lo, hi, (4_000_000,), generator=g, dtype=torch.int64)
        off = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(lens, 0)]).to(dev)
        ok = case(name, off, lens.numel()) and ok
print("all outputs equal the input" if ok else "FAILED")
sys.exit(0 if ok else 1)
