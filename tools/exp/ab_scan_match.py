#!/usr/bin/env python3
"""The query's side of the matrix product, for the scan (distance bytes) and the fused count: 1.0 on the channels that DIFFER from the query's base
(scan_mfma_match 0: three channels of four are non-zero, the product counts mismatches) against -1.0 on the one channel that EQUALS it, counted down
from k (1: a third of the non-zero entries -- the matrix pipe's power is what lowers the clock in a queue from idle, profiles/r05_ablate_count_parts.txt).
Same kernels, only the host-built table and the accumulators' start values differ.  Evidence build.  Both forms are first checked against the oracle
(every k, sizes around rounds / trips, thresholds on both sides of k, data where most windows hit) and against each other at 10^9 bases."""
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
outs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
ctx.sync()
oracle_py.lib()
rng = np.random.default_rng(12)
ALPHA8 = np.frombuffer(b"ACGTacgt", dtype=np.uint8)
ok = True
for match in (0, 1):
    ctx.require_variant("scan_mfma_match", match)
    for kk in (1, 2, 15, 16, 17, 31, 32):
        for nn in (kk, 1055, 1056, 1057, 2080, 2081, 4128, 4129, 5153, 9 * 1024 + 77, 200003):
            if nn < kk:
                continue
            qq = int(rng.integers(0, 1 << 62)) | (int(rng.integers(0, 4)) << 62)
            for kind in ("random", "periodic"):
                if kind == "random":
                    s_ = ALPHA8[rng.integers(0, 8, size=nn)]
                else:
                    unit = np.array([ord("ACGT"[(qq >> (2 * i)) & 3]) for i in range(kk)], dtype=np.uint8)
                    s_ = np.tile(unit, nn // kk + 1)[:nn].copy()
                    s_[rng.integers(0, nn, size=max(1, nn // 50))] = ord("a")
                t_ = torch.from_numpy(s_).to(dev)
                want = oracle_py.kmer_hdist_scan(s_, kk, qq)
                d_ = torch.full((nn,), 0xEE, dtype=torch.uint8, device=dev)
                torch.cuda.synchronize()
                ctx.kmer_hdist_scan_dev(t_, nn, kk, qq, d_)
                ctx.sync()
                got = d_.cpu().numpy()
                if not (np.array_equal(got[:nn - kk + 1], want) and bool((got[nn - kk + 1:] == 0xEE).all())):
                    print(f"SCAN MISMATCH match {match} {kind} k {kk} n {nn}")
                    ok = False
                for tau in sorted({0, 1, kk // 2, max(kk - 1, 0), kk, kk + 1, 31, 32, 33, 2**32 - 1}):
                    ctx.kmer_hdist_count_dev(t_, nn, kk, qq, tau, cnt)
                    ctx.kmer_hdist_count_dev(t_, nn, kk, qq, tau, cnt)
                    ctx.sync()
                    if int(cnt.item()) != int((want <= tau).sum()):
                        print(f"COUNT MISMATCH match {match} {kind} k {kk} n {nn} tau {tau}: {int(cnt.item())} != {int((want <= tau).sum())}")
                        ok = False
print("small sizes vs oracle (scan and count, both table forms):", "ok" if ok else "FAILED", flush=True)
base = None
for match in (0, 1):
    ctx.require_variant("scan_mfma_match", match)
    d = torch.full((n,), 0xEE, dtype=torch.uint8, device=dev)
    ctx.kmer_hdist_scan_dev(ref, n, k, q, d)
    ctx.sync()
    if base is None:
        base = d
    else:
        same = torch.equal(d, base)
        ok &= same
        print("10^9 bases: distance bytes of the two forms identical:", same, flush=True)
    for tau in (8, 18, 23, 31):
        ctx.kmer_hdist_count_dev(ref, n, k, q, tau, cnt)
        ctx.sync()
        want = int((base[:n - k + 1] <= tau).sum().item())
        if int(cnt.item()) != want:
            print(f"10^9 COUNT MISMATCH match {match} tau {tau}: {int(cnt.item())} != {want}")
            ok = False
del d
flip = [0]


def scan():
    flip[0] ^= 1
    ctx.kmer_hdist_scan_dev(ref, n, k, q, outs[flip[0]])


def count():
    ctx.kmer_hdist_count_dev(ref, n, k, q, 8, cnt)


def burst(f, B=8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f()
    a.record(stream)
    for _ in range(B):
        f()
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / B


def queue(f, N=96, every=8):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N // every + 1)]
    torch.cuda.synchronize()
    time.sleep(1.0)
    ev[0].record(stream)
    for i in range(N):
        f()
        if (i + 1) % every == 0:
            ev[(i + 1) // every].record(stream)
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) * 1e3 / every for i in range(N // every)]


for name, f, alg in (("scan ", scan, 2 * (n - k + 1)), ("count", count, n - k + 1)):
    res = {0: [], 1: []}
    for rnd in range(7):
        for match in (0, 1):
            ctx.require_variant("scan_mfma_match", match)
            t = burst(f)
            if rnd:
                res[match].append(t)
    for rep in range(3):
        for match in (0, 1):
            ctx.require_variant("scan_mfma_match", match)
            g = queue(f)
            m = statistics.median(res[match])
            print(f"{name} {'matches counted down from k' if match else 'mismatches counted (ships) '}: bursts {m*1e3:6.1f} us ({alg/m/8e7:4.1f} %)   from idle: mean {sum(g)/len(g):6.1f} us ({alg/(sum(g)/len(g))/8e4:4.1f} %), settled {sum(g[-2:])/2:6.1f}, slowest group {max(g):6.1f}   groups: {' '.join(f'{x:.0f}' for x in g)}", flush=True)
ctx.require_variant("scan_mfma_match", 0)
sys.exit(0 if ok else 1)
