#!/usr/bin/env python3
"""In-process A/B of a context knob on the config-5 scan (default: scan_v2 0 vs 1)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd

key = sys.argv[1] if len(sys.argv) > 1 else "scan_v2"
values = [int(v) for v in sys.argv[2:]] or [0, 1]
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream)
n, k = 10**9, 31
ref = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(ref, n, 1)
dist = {v: torch.empty(n - k + 1, dtype=torch.uint8, device=dev) for v in values}
q = 0x1B1B1B1B1B1B1B1B & ((1 << 62) - 1)
res = {v: [] for v in values}
for rnd in range(9):
    for v in values:
        ctx.set_variant(key, v)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
        ev[0].record(stream)
        for i in range(6):
            ctx.kmer_hdist_scan_dev(ref, n, k, q, dist[v])
            ev[i + 1].record(stream)
        torch.cuda.synchronize()
        res[v].append(statistics.mean(ev[i].elapsed_time(ev[i + 1]) for i in range(2, 6)))
ctx.sync()
for v in values[1:]:
    assert torch.equal(dist[values[0]], dist[v])
for v in values:
    ms = statistics.median(res[v])
    print(f"{key}={v}: {ms:.4f} ms  {2 * (n - k + 1) / ms / 1e6:.0f} GB/s", flush=True)
