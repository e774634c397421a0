#!/usr/bin/env python3
"""Host-visible time of bitnuc_batch_plan_build_dev and bitnuc_batch_word_offsets_dev (both synchronous) for L-base reads."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd

dev = torch.device("cuda:0")
ctx = bitnuc_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
for L in [int(a) for a in sys.argv[1:]] or [150, 32, 1000]:
    count = 10**9 // L
    off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L
    wo = torch.empty(count + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    plan = bitnuc_amd.BatchPlan(ctx)
    tb, tw = [], []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan.build(off, count)
        tb.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter()
        total = ctx.batch_word_offsets_dev(off, count, wo)
        tw.append((time.perf_counter() - t0) * 1e3)
    assert plan.total_words == total
    print(f"L={L}: {count} sequences: plan build {min(tb[1:]):.3f} ms, word offsets {min(tw[1:]):.3f} ms (host-visible, synchronous, best of 5)", flush=True)
    plan.close()
