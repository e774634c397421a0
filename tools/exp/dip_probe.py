#!/usr/bin/env python3
"""What slows down in the dip that follows the first launches of a matrix-core queue on an idle chip: the matrix-core kernel only (its cycles), or the memory
system as a whole?  A queue of fused-count launches (bitnuc_kmer_hdist_count_dev, 10^9 bases) from an idle chip, with one launch of a kernel WITHOUT matrix
instructions timed after every fourth count launch: packed-vs-packed hdist (read-only, 0.5 GB, at 0.875 of 8 TB/s on its own) in one run, bulk encode of 10^9 bases
(1 B read + 0.25 B written per base, 0.78) in another.  Product library.  Printed: per group, the count's time per launch and the probe's time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bitnuc_amd

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream)
n, k = 10**9, 31
q = 0x1B1B1B1B1B1B1B1B & ((1 << 62) - 1)
ref = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(ref, n, 0xB17C0DE)
nw = (n + 31) // 32
wa = torch.empty(nw, dtype=torch.int64, device=dev)
wb = torch.empty(nw, dtype=torch.int64, device=dev)
ctx.encode_dev(ref, n, wa)
other = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(other, n, 0xB17C0DE + 200)
ctx.encode_dev(other, n, wb)
del other
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
res = torch.zeros(1, dtype=torch.int32, device=dev)
wout = torch.empty(nw, dtype=torch.int64, device=dev)
ctx.sync()
PROBES = {"hdist of two packed 10^9-base buffers (read-only)": lambda: ctx.hdist_dev(wa, nw, wb, nw, n, res),
          "bulk encode of 10^9 bases": lambda: ctx.encode_dev(ref, n, wout)}


def alone(f, reps=24):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f()
    a.record(stream)
    for _ in range(reps):
        f()
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for name, probe in PROBES.items():
    base = alone(probe)
    G = 24  # groups of (4 count launches + 1 probe)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * G + 1)]
    torch.cuda.synchronize()
    time.sleep(1.0)
    ev[0].record(stream)
    for g in range(G):
        for _ in range(4):
            ctx.kmer_hdist_count_dev(ref, n, k, q, 8, cnt)
        ev[2 * g + 1].record(stream)
        probe()
        ev[2 * g + 2].record(stream)
    torch.cuda.synchronize()
    c = [ev[2 * g].elapsed_time(ev[2 * g + 1]) * 1e3 / 4 for g in range(G)]
    p = [ev[2 * g + 1].elapsed_time(ev[2 * g + 2]) * 1e3 for g in range(G)]
    print(f"probe: {name}; alone, sustained: {base:.1f} us")
    print("  count, us per launch :", " ".join(f"{x:.0f}" for x in c))
    print("  probe, us            :", " ".join(f"{x:.0f}" for x in p))
    worst = max(range(G), key=lambda g: c[g])
    print(f"  count at its slowest (group {worst}): {c[worst]:.0f} us = {c[worst] / min(c):.2f} x its fastest; the probe there: {p[worst]:.0f} us = {p[worst] / min(p):.2f} x its fastest, {p[worst] / base:.2f} x alone", flush=True)
