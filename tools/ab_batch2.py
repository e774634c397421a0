#!/usr/bin/env python3
"""In-process A/B of the two ragged-batch formulations: the table-driven kernels (encode_batch_dev / decode_batch_dev: tile
records + O(1) pad-scatter lookup; the record pre-kernel is inside the timing) and the layout plan (BatchPlan).  Interleaved
rounds, median ms per launch, 10^9 bases."""
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
back = torch.empty(N, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(seq, N, 0xB17C0DE)
ctx.sync()
impls = ["tables"]


def once(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    fn()
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b)


for L in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "32,100,150,151,250,1000,10000,1000000".split(","))]:
    count = N // L
    if L == 151:  # ragged: lengths 100..250 around 150
        g = torch.Generator(device="cpu").manual_seed(5)
        lens = torch.randint(100, 251, (N // 180,), generator=g, dtype=torch.int64)
        off = torch.zeros(lens.numel() + 1, dtype=torch.int64)
        off[1:] = torch.cumsum(lens, 0)
        off = off.to(dev)
        count = lens.numel()
    else:
        off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L
    wo = torch.empty(count + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    total = ctx.batch_word_offsets_dev(off, count, wo)
    words = torch.empty(total, dtype=torch.int64, device=dev)
    res = {i: ([], []) for i in impls}
    res["plan"] = ([], [])
    import time
    plan = bitnuc_amd.BatchPlan(ctx)
    tb = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan.build(off, count)
        tb.append((time.perf_counter() - t0) * 1e3)
    assert plan.total_words == total
    for rnd in range(9):
        for i in impls:
            e = once(lambda: ctx.encode_batch_dev(seq, off, wo, count, total, words))
            d = once(lambda: ctx.decode_batch_dev(words, wo, off, count, total, back))
            if rnd >= 2:
                res[i][0].append(e)
                res[i][1].append(d)
        e = once(lambda: plan.encode_dev(seq, words))
        d = once(lambda: plan.decode_dev(words, back))
        if rnd >= 2:
            res["plan"][0].append(e)
            res["plan"][1].append(d)
    ctx.sync()
    plan.close()
    nb = int(off[-1].item())
    ok = bool(torch.equal(seq[:nb], back[:nb]))
    alg = nb + 8 * total
    line = f"L={L:8d} count={count:9d}"
    for i in impls + ["plan"]:
        e, d = statistics.median(res[i][0]), statistics.median(res[i][1])
        line += f" | {i}: enc {e:.4f} ms {alg/e/1e6:5.0f} GB/s  dec {d:.4f} ms {alg/d/1e6:5.0f} GB/s"
    print(line + f" | plan build {min(tb[1:]):.3f} ms (host-synchronous, incl. word offsets) | roundtrip {'ok' if ok else 'MISMATCH'}", flush=True)
