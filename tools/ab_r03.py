#!/usr/bin/env python3
"""Round-3 A/B in one process (sustained bursts, alternating buffer sets, interleaved rounds):
  batch    150-base reads, 10^9 bases: plan kernels | table-driven with asynchronous plan emission (product) | round 2's
           table-driven kernels (evidence build: block_owner + batch2), encode and decode; also the emit kernel alone
  windows  every 31-base window of 10^9 bases: strip kernel (rounds of 992) vs line-aligned kernel, 1 / 2 / 4 rounds per trip
usage: ab_r03.py [batch] [windows] [L=150]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd
from bitnuc_amd import build as _build

what = [a for a in sys.argv[1:] if "=" not in a] or ["batch", "windows"]
kv = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
prod = bitnuc_amd.Context(0, stream=stream.cuda_stream)
evid = bitnuc_amd.Context(0, stream=stream.cuda_stream, lib_path=_build.ensure_built(sweep=True))
N = 10**9
BURST, ROUNDS = 8, 5


def sustained(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(0)
    fn(1)
    a.record(stream)
    for i in range(BURST):
        fn(i)
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / BURST


def table(title, cases, alg):
    res = {k: [] for k in cases}
    for rnd in range(ROUNDS + 1):
        for k, fn in cases.items():
            t = sustained(fn)
            if rnd:
                res[k].append(t)
    print(title)
    for k, v in res.items():
        m = statistics.median(v)
        print(f"  {k:58s} {m:.4f} ms  {alg / m / 1e6:6.0f} GB/s  {alg / m / 1e6 / 8000:.3f}   (min {min(v):.4f} max {max(v):.4f})")


seq = torch.empty(N, dtype=torch.uint8, device=dev)
prod.nucgen_dev(seq, N, 0xB17C0DE)
prod.sync()

if "batch" in what:
    L = int(kv.get("L", 150))
    count = N // L
    off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L
    wo = torch.empty(count + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    total = prod.batch_word_offsets_dev(off, count, wo)
    ws = [torch.empty(total, dtype=torch.int64, device=dev) for _ in range(2)]
    bs = [torch.empty(N, dtype=torch.uint8, device=dev) for _ in range(2)]
    plan = bitnuc_amd.BatchPlan(prod, off, count)
    plan.encode_dev(seq, ws[0])
    plan.encode_dev(seq, ws[1])
    prod.sync()
    ref = ws[0].clone()
    alg = L * count + 8 * total
    evid.require_variant("batch_tables_impl", 0)
    table(f"encode, {count} reads of {L} bases ({alg / 1e9:.3f} GB algorithmic; tables not counted)", {
        "plan kernel (plan built once)": lambda i: plan.encode_dev(seq, ws[i & 1]),
        "tables: plan_emit_kernel + plan kernel (product)": lambda i: prod.encode_batch_dev(seq, off, wo, count, total, ws[i & 1]),
        "tables, round 2: block_owner + encode_batch2 (evidence)": lambda i: evid.encode_batch_dev(seq, off, wo, count, total, ws[i & 1]),
    }, alg)
    prod.sync()
    ok_e = bool(torch.equal(ws[0], ref) and torch.equal(ws[1], ref))
    table("decode", {
        "plan kernel (plan built once)": lambda i: plan.decode_dev(ws[i & 1], bs[i & 1]),
        "tables: plan_emit_kernel + plan kernel (product)": lambda i: prod.decode_batch_dev(ws[i & 1], wo, off, count, total, bs[i & 1]),
        "tables, round 2: block_owner + decode_batch2 (evidence)": lambda i: evid.decode_batch_dev(ws[i & 1], wo, off, count, total, bs[i & 1]),
    }, alg)
    prod.sync()
    print("  outputs:", "same words" if ok_e else "WORD MISMATCH", "/", "round trip ok" if torch.equal(bs[0][:L * count], seq[:L * count]) and torch.equal(bs[1][:L * count], seq[:L * count]) else "ROUND TRIP MISMATCH")
    plan.close()
    del ws, bs, ref

if "windows" in what:
    k = 31
    nwin = N - k + 1
    outs = [torch.empty(nwin, dtype=torch.int64, device=dev) for _ in range(2)]
    alg = N + 8 * nwin
    cases = {}
    for impl in (0, 1):
        for u in (1, 2, 4):
            def fn(i, impl=impl, u=u):
                evid.require_variant("slide_impl", impl)
                evid.require_variant("slide_rounds", u)
                evid.require_variant("slide2_rounds", u)
                evid.as_2bit_batch_dev(seq, k, 1, nwin, outs[i & 1])
            cases[f"{'line-aligned, computed in place' if impl else 'strip kernel (rounds of 992)'}, {u} round(s) per trip"] = fn
    table(f"every {k}-base window of 10^9 bases ({alg / 1e9:.3f} GB algorithmic)", cases, alg)
    evid.require_variant("slide_impl", 0)
    evid.require_variant("slide_rounds", 1)
    evid.as_2bit_batch_dev(seq, k, 1, nwin, outs[0])
    evid.require_variant("slide_impl", 1)
    evid.as_2bit_batch_dev(seq, k, 1, nwin, outs[1])
    evid.sync()
    print("  outputs:", "same" if torch.equal(outs[0], outs[1]) else "MISMATCH")
