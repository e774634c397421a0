#!/usr/bin/env python3
"""encode_batch / decode_batch throughput vs read length (10^9 bases total)."""
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


def timed(fn, reps=8):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record(stream)
    for i in range(reps):
        fn()
        ev[i + 1].record(stream)
    torch.cuda.synchronize()
    return statistics.median(ev[i].elapsed_time(ev[i + 1]) for i in range(2, reps))


for L in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "32,100,128,150,151,160,250,1000,10000,1000000".split(","))]:
    count = N // L
    off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L
    wo = torch.empty(count + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    import time
    total = ctx.batch_word_offsets_dev(off, count, wo)
    t0 = time.perf_counter()
    for _ in range(5):
        total = ctx.batch_word_offsets_dev(off, count, wo)
    t_off = (time.perf_counter() - t0) / 5 * 1e3
    words = torch.empty(total, dtype=torch.int64, device=dev)
    e = timed(lambda: ctx.encode_batch_dev(seq, off, wo, count, total, words))
    d = timed(lambda: ctx.decode_batch_dev(words, wo, off, count, total, back))
    ef = timed(lambda: ctx.encode_fixed_dev(seq, L, L, count, words)) if total == count * ((L + 31) // 32) else 0
    df = timed(lambda: ctx.decode_fixed_dev(words, L, L, count, back))
    ctx.sync()
    nb = L * count
    ok = bool(torch.equal(seq[:nb], back[:nb]))
    alg = nb + 8 * total
    print(f"L={L:8d} count={count:9d} encode {e:.4f} ms {alg/e/1e6:6.0f} GB/s | decode {d:.4f} ms {alg/d/1e6:6.0f} GB/s | fixed: encode {ef:.4f} ms {alg/ef/1e6:6.0f} GB/s decode {df:.4f} ms {alg/df/1e6:6.0f} GB/s | word_offsets {t_off:.3f} ms (host-synchronous) | roundtrip {'ok' if ok else 'MISMATCH'}", flush=True)
