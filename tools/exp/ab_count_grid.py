#!/usr/bin/env python3
"""Fused count (bitnuc_kmer_hdist_count_dev) of the matrix-core scan: a resident grid with one accumulator + ticket (scan_mfma_count_persist 1)
against one trip per wave with spread partial accumulators + a finishing launch (0), trips of 4 and 2 rounds -- evidence build; counts checked
against the distance bytes.  Bursts of 8 and a 96-launch queue from an idle chip."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
d = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.kmer_hdist_scan_dev(ref, n, k, q, d)
ctx.sync()
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
# the count's own tiling at sizes around its rounds, every k, against the oracle's distance bytes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import oracle_py
oracle_py.lib()
rng = np.random.default_rng(9)
small_ok = True
ctx.require_variant("scan_mfma_count_form", 1)
for U in (4, 3, 2):
    ctx.require_variant("scan_mfma_count_rounds", U)
    for kk in (1, 2, 15, 16, 17, 31, 32):
        for nn in (kk, 1055, 1056, 1057, 2080, 2081, 4128, 4129, 5153, 9 * 1024 + 77, 200003):
            if nn < kk:
                continue
            s_ = np.frombuffer(b"ACGTacgt", dtype=np.uint8)[rng.integers(0, 8, size=nn)]
            qq = int(rng.integers(0, 1 << 62)) | (int(rng.integers(0, 4)) << 62)
            t_ = torch.from_numpy(s_).to(dev)
            want_d = oracle_py.kmer_hdist_scan(s_, kk, qq)
            for tau in (0, kk // 2, kk):
                torch.cuda.synchronize()
                ctx.kmer_hdist_count_dev(t_, nn, kk, qq, tau, cnt)
                ctx.sync()
                if int(cnt.item()) != int((want_d <= tau).sum()):
                    print(f"SMALL MISMATCH own tiling U {U} k {kk} n {nn} tau {tau}: {int(cnt.item())} != {int((want_d <= tau).sum())}")
                    small_ok = False
    bad = np.frombuffer(b"ACGTacgt", dtype=np.uint8)[rng.integers(0, 8, size=50000)].copy()
    for pos in (0, 1023, 1024, 1040, 1055, 4096, 4097, 30000, 49999):
        b2 = bad.copy()
        b2[pos] = ord("N")
        tb = torch.from_numpy(b2).to(dev)
        torch.cuda.synchronize()
        ctx.kmer_hdist_count_dev(tb, 50000, 31, 0, 3, cnt)
        try:
            ctx.sync()
            print(f"own tiling: invalid byte at {pos} NOT reported")
            small_ok = False
        except bitnuc_amd.NucleotideError as e:
            if (e.byte, e.index) != (ord("N"), pos):
                print(f"own tiling: invalid byte at {pos} reported as {(e.byte, e.index)}")
                small_ok = False
print("own tiling, small sizes vs oracle:", "ok" if small_ok else "FAILED", flush=True)
KEYS = ("scan_mfma_shift", "scan_mfma_unroll", "scan_mfma_count_rounds", "scan_mfma_count_persist", "scan_mfma_count_form", "scan_mfma_grid", "scan_mfma_count_grid")


def knob_values(f):
    sh, U, cp, cf = f[:4]
    g = f[4] if len(f) > 4 else 4
    return (sh, 4 if U == 3 else U, U, cp, cf, g, g)  # natural tiling: scan_mfma_unroll / scan_mfma_grid; own tiling: scan_mfma_count_rounds / scan_mfma_count_grid


FORMS = [(4, 4, 1, 0), (4, 3, 1, 1, 18), (4, 3, 1, 1, 6), (4, 3, 1, 1, 24), (4, 4, 1, 1, 4), (4, 4, 1, 1, 12), (4, 2, 1, 1, 16)]  # (scan shift, rounds per trip, resident grid, count_form[, workgroups per CU])  # (shift, rounds per trip, resident grid, count_form)
ok = True
for tau in (18, 8, 31):
    want = int((d[:n - k + 1] <= tau).sum().item())
    for f in FORMS:
        sh, U, cp, cf = f[:4]
        for key, v in zip(KEYS, knob_values(f)):
            ctx.require_variant(key, v)
        for _ in range(2):  # twice: the accumulators must be zero again after a call
            ctx.kmer_hdist_count_dev(ref, n, k, q, tau, cnt)
            ctx.sync()
            if int(cnt.item()) != want:
                print(f"MISMATCH shift {sh} U {U} count_persist {cp} form {cf} tau {tau}: {int(cnt.item())} != {want}")
                ok = False
print("counts:", "ok" if ok else "FAILED", flush=True)


def burst(B=8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx.kmer_hdist_count_dev(ref, n, k, q, 18, cnt)
    a.record(stream)
    for _ in range(B):
        ctx.kmer_hdist_count_dev(ref, n, k, q, 18, cnt)
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / B


def queue(N=96):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    torch.cuda.synchronize()
    time.sleep(1.0)
    ev[0].record(stream)
    for i in range(N):
        ctx.kmer_hdist_count_dev(ref, n, k, q, 18, cnt)
        ev[i + 1].record(stream)
    torch.cuda.synchronize()
    us = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(N)]
    return sum(us) / N, sum(us[-16:]) / 16, max(us[1:])



res = {f: [] for f in FORMS}
for rnd in range(6):
    for f in FORMS:
        for key, v in zip(KEYS, knob_values(f)):
            ctx.require_variant(key, v)
        t = burst()
        if rnd:
            res[f].append(t)
for f in FORMS:
    m = statistics.median(res[f])
    for key, v in zip(KEYS, knob_values(f)):
        ctx.require_variant(key, v)
    mean, settled, worst = queue()
    print(f"{'own tiling (4 MFMA)   ' if f[3] else 'natural tiling shift ' + str(f[0])} U {f[1]} grid {f[4] if len(f) > 4 else 4}/CU {'resident grid + ticket   ' if f[2] else 'one trip per wave + finish'}: bursts {m*1e3:6.1f} us ({(n-k+1)/m/8e7:4.1f} % of 8 TB/s)   from idle: mean of 96 {mean:6.1f} us, last 16 {settled:6.1f} us, slowest {worst:6.1f} us", flush=True)
sys.exit(0 if ok and small_ok else 1)
