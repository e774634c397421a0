#!/usr/bin/env python3
"""Fused count: four channels per base (scan_mfma_count_form 1: kmer_count_mfma_kernel, four MFMAs per 1024 windows, ships) against three (2: kmer_count3_mfma_kernel,
[b != q] affine in an (A, C, G) one-hot with T = 0: three MFMAs, three ds_read_b128, six more vector instructions for the G-nibble packing) x trips of 2 / 3 / 4
rounds x workgroups per CU.  Why: the matrix pipe's power is what lowers the clock in a queue from idle (profiles/r05_ablate_count_parts.txt).  Evidence build.  Every
form is first checked against the oracle (small sizes, every k, thresholds on both sides of k, data where every window hits, invalid bytes) and against the
distance bytes at 10^9 bases."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import bitnuc_amd
import oracle_py
from bitnuc_amd import build

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream, lib_path=build.ensure_built(sweep=True))
n, k = 10**9, 31
q = 0x1B1B1B1B1B1B1B1B & ((1 << 62) - 1)
ref = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(ref, n, 0xB17C0DE)
d = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.kmer_hdist_scan_dev(ref, n, k, q, d)
ctx.sync()
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
oracle_py.lib()
rng = np.random.default_rng(10)
ALPHA8 = np.frombuffer(b"ACGTacgt", dtype=np.uint8)
ctx.require_variant("scan_mfma_count_emit", 2)

FORMS = [(1, 3, 18), (2, 3, 18), (2, 3, 24), (2, 4, 12), (2, 4, 18), (2, 2, 24)]  # (count_form, rounds per trip, workgroups per CU[, 1 = a wave owns a contiguous run of trips and copies the halo's operands from the strip])
if len(sys.argv) > 1:
    FORMS = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]


def chained(f):
    return " chained" if len(f) > 3 and f[3] else ""


def use(f):
    ctx.require_variant("scan_mfma_count_form", f[0])
    ctx.require_variant("scan_mfma_count_rounds", f[1])
    ctx.require_variant("scan_mfma_count_grid", f[2])
    if len(f) > 3 and f[3]:
        ctx.require_variant("scan_mfma_count_chain", 1)  # (the chained form lives in commit 7c7c8a3 only: it lost its A/B, profiles/r05_ab_count_chain.txt)


small_ok = True
for f in sorted({(f[0], f[1], 18, f[3] if len(f) > 3 else 0) for f in FORMS}) + sorted({(f[0], f[1], 1, 1) for f in FORMS if len(f) > 3 and f[3]}):
    use(f)
    for kk in (1, 2, 15, 16, 17, 31, 32):
        for nn in (kk, 1055, 1056, 1057, 2080, 2081, 4128, 4129, 5153, 9 * 1024 + 77, 200003):
            if nn < kk:
                continue
            qq = int(rng.integers(0, 1 << 62)) | (int(rng.integers(0, 4)) << 62)
            for kind in ("random", "periodic"):
                if kind == "random":
                    s_ = ALPHA8[rng.integers(0, 8, size=nn)]
                else:  # the query's own bases repeated with a few substitutions: most windows at a multiple of k are near hits
                    unit = np.array([ord("ACGT"[(qq >> (2 * i)) & 3]) for i in range(kk)], dtype=np.uint8)
                    s_ = np.tile(unit, nn // kk + 1)[:nn].copy()
                    s_[rng.integers(0, nn, size=max(1, nn // 50))] = ord("A")
                t_ = torch.from_numpy(s_).to(dev)
                want_d = oracle_py.kmer_hdist_scan(s_, kk, qq)
                for tau in sorted({0, 1, kk // 2, max(kk - 1, 0), kk, kk + 1, 31, 32, 33, 1000, 2**32 - 1}):
                    torch.cuda.synchronize()
                    ctx.kmer_hdist_count_dev(t_, nn, kk, qq, tau, cnt)
                    ctx.kmer_hdist_count_dev(t_, nn, kk, qq, tau, cnt)
                    ctx.sync()
                    if int(cnt.item()) != int((want_d <= tau).sum()):
                        print(f"SMALL MISMATCH form {f} U {f[1]} {kind} k {kk} n {nn} tau {tau}: {int(cnt.item())} != {int((want_d <= tau).sum())}")
                        small_ok = False
    bad = ALPHA8[rng.integers(0, 8, size=50000)].copy()
    for pos in (0, 15, 16, 1023, 1024, 1040, 1055, 1056, 3071, 3072, 3104, 4096, 4097, 30000, 49999):
        b2 = bad.copy()
        b2[pos] = ord("N")
        if pos + 9 < b2.size:
            b2[pos + 9] = ord("X")
        tb = torch.from_numpy(b2).to(dev)
        torch.cuda.synchronize()
        ctx.kmer_hdist_count_dev(tb, 50000, 31, 0, 3, cnt)
        try:
            ctx.sync()
            print(f"form {f[0]} U {f[1]}: invalid byte at {pos} NOT reported")
            small_ok = False
        except bitnuc_amd.NucleotideError as e:
            if (e.byte, e.index) != (ord("N"), pos):
                print(f"form {f[0]} U {f[1]}: invalid byte at {pos} reported as {(e.byte, e.index)}")
                small_ok = False
print("small sizes vs oracle:", "ok" if small_ok else "FAILED", flush=True)

ok = True
for tau in (18, 8, 23, 31):
    want = int((d[:n - k + 1] <= tau).sum().item())
    for f in FORMS:
        use(f)
        for _ in range(2):  # twice: the accumulator must be zero again after a call
            ctx.kmer_hdist_count_dev(ref, n, k, q, tau, cnt)
            ctx.sync()
            if int(cnt.item()) != want:
                print(f"MISMATCH form {f} U {f[1]} grid {f[2]} tau {tau}: {int(cnt.item())} != {want}")
                ok = False
print("counts at 10^9 bases:", "ok" if ok else "FAILED", flush=True)
TAU = 8


def burst(B=8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx.kmer_hdist_count_dev(ref, n, k, q, TAU, cnt)
    a.record(stream)
    for _ in range(B):
        ctx.kmer_hdist_count_dev(ref, n, k, q, TAU, cnt)
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / B


def queue(N=96):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N // 8 + 1)]
    torch.cuda.synchronize()
    time.sleep(1.0)
    ev[0].record(stream)
    for i in range(N):
        ctx.kmer_hdist_count_dev(ref, n, k, q, TAU, cnt)
        if i % 8 == 7:
            ev[i // 8 + 1].record(stream)
    torch.cuda.synchronize()
    us = [ev[i].elapsed_time(ev[i + 1]) * 1e3 / 8 for i in range(N // 8)]
    return sum(us) / len(us), us[0], sum(us[-2:]) / 2, max(us)


res = {f: [] for f in FORMS}
for rnd in range(6):
    for f in FORMS:
        use(f)
        t = burst()
        if rnd:
            res[f].append(t)
for f in FORMS:
    m = statistics.median(res[f])
    use(f)
    mean, first, settled, worst = sorted(queue() for _ in range(3))[1]  # the median of three queues, each from an idle chip
    print(f"form {f[0]}{chained(f)} trips of {f[1]} grid {f[2]:2d}/CU: bursts {m*1e3:6.1f} us ({(n-k+1)/m/8e7:4.1f} % of 8 TB/s)   from idle (groups of 8): mean of 96 {mean:6.1f} us, first 8 {first:6.1f}, last 16 {settled:6.1f}, slowest group {worst:6.1f} us", flush=True)
sys.exit(0 if ok and small_ok else 1)
