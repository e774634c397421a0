#!/usr/bin/env python3
"""Config-5 scan: kmer_scan2_kernel GEN 1 (scan_impl 1, ships since round 4: two-LUT plane build + scalar halo, 9.4 VALU per window) against GEN 0
(scan_impl 6: rounds 2-3's plane build, 10.2 VALU per window) and kmer_scan3_kernel (scan_impl 2 / 3 / 4 / 5: a wave owns 12 / 20 / 16 / 32 consecutive rounds; plane
build without the v_and pair, halo planes carried between trips) -- evidence build, one process.  The scan is VALU-issue bound and the
chip lowers its clock under such a kernel (profiles/r04_launch_series.txt), so three readings per form, interleaved:
  bursts   sustained bursts of 8 launches between host syncs, median of 6 rounds (how bench.py times the side blocks)
  queue    64 launches in ONE queue: mean of all, and mean of the last 16 (the settled clock)
Outputs are compared word for word between the forms (10^9 bases, k = 31) and, for k in {1, 16, 17, 31, 32}, on 3 * 10^6 + 77 bases."""
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
n, k = 10**9, 31
q = 0x1B1B1B1B1B1B1B1B & ((1 << 62) - 1)
ref = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(ref, n, 0xB17C0DE)
dists = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
ctx.sync()
IMPLS = (1, 6, 2, 4)
NAMES = {1: "scan2 GEN1 (2 LUTs, scalar halo: ships)", 6: "scan2 GEN0 (rounds 2-3)", 2: "scan3, 12 rounds/wave", 3: "scan3, 20 rounds/wave", 4: "scan3, 16 rounds/wave", 5: "scan3, 32 rounds/wave"}

# ---- same output ----
ok = True
outs = {}
for impl in IMPLS:
    ctx.require_variant("scan_impl", impl)
    d = torch.full((n,), 0xEE, dtype=torch.uint8, device=dev)
    ctx.kmer_hdist_scan_dev(ref, n, k, q, d)
    ctx.sync()
    outs[impl] = d
for impl in IMPLS[1:]:
    same = torch.equal(outs[impl], outs[1])
    ok = ok and same
    print(f"10^9 bases, k=31: impl {impl} == impl 1: {same}; bytes past the last window untouched: {bool((outs[impl][n - k + 1:] == 0xEE).all())}", flush=True)
del outs
small = 3 * 10**6 + 77
for kk in (1, 16, 17, 31, 32):
    qq = (0x2B1B4E1B1B1B1B1B ^ (kk * 0x9E3779B97F4A7C15)) & ((1 << 64) - 1)
    res = []
    for impl in IMPLS:
        ctx.require_variant("scan_impl", impl)
        d = torch.zeros(small, dtype=torch.uint8, device=dev)
        ctx.kmer_hdist_scan_dev(ref, small, kk, qq, d)
        ctx.sync()
        res.append(d)
    same = all(torch.equal(r, res[0]) for r in res[1:])
    ok = ok and same
    print(f"{small} bases, k={kk}: all forms agree: {same}", flush=True)
flip = [0]


def scan():
    flip[0] ^= 1
    ctx.kmer_hdist_scan_dev(ref, n, k, q, dists[flip[0]])


def burst(B=8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    scan()
    a.record(stream)
    for _ in range(B):
        scan()
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / B


def queue(N=64):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    scan()
    ev[0].record(stream)
    for i in range(N):
        scan()
        ev[i + 1].record(stream)
    torch.cuda.synchronize()
    us = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(N)]
    return sum(us) / N, sum(us[-16:]) / 16, max(us)


res = {i: [] for i in IMPLS}
for rnd in range(7):
    for impl in IMPLS:
        ctx.require_variant("scan_impl", impl)
        t = burst()
        if rnd:
            res[impl].append(t)
alg = 2 * (n - k + 1)
for impl in IMPLS:
    m = statistics.median(res[impl])
    print(f"bursts  {NAMES[impl]:40s} {m*1e3:7.1f} us  {alg/m/1e6:6.0f} GB/s  {alg/m/8e7:5.1f} % of 8 TB/s", flush=True)
for rep in range(2):
    for impl in IMPLS:
        ctx.require_variant("scan_impl", impl)
        import time
        time.sleep(0.3)
        mean, settled, worst = queue()
        print(f"queue{rep} {NAMES[impl]:40s} mean of 64 {mean:6.1f} us ({alg/mean/8e4:4.1f} %)  last 16 {settled:6.1f} us ({alg/settled/8e4:4.1f} %)  slowest {worst:6.1f} us", flush=True)
ctx.require_variant("scan_impl", 1)
print("all outputs equal" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
