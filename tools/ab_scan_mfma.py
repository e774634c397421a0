#!/usr/bin/env python3
"""Config-5 scan: kmer_scan2_kernel GEN 1 (scan_impl 1: bit-planes + v_alignbit + v_bcnt, 9.4 vector instructions per window) against
kmer_scan_mfma_kernel (scan_impl 7: the one-hot contraction on the matrix cores, csrc/evidence/scan_mfma_evidence.h: the natural-layout tiling that shipped first) -- evidence build, one process.
Correctness first: impl 7 (every pack mode / trip length) against the oracle on sizes around the rounds for k in {1, 2, 15, 16, 17, 31, 32},
against impl 1 on 10^9 bases, the fused count, the first invalid byte.  Then three timing readings per form, interleaved (the bit-plane
scan is VALU-issue bound and follows the chip's clock, profiles/r04_clock_series.txt):
  bursts   sustained bursts of 8 launches between host syncs, median of 6 rounds (how bench.py times the side blocks)
  queue    96 launches in ONE queue from an idle chip: mean of all (the conservative reading), mean of the last 16, the slowest
usage: ab_scan_mfma.py [quick]"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import bitnuc_amd
from bitnuc_amd import build
import oracle_py

quick = "quick" in sys.argv
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream, lib_path=build.ensure_built(sweep=True))
assert ctx.get("sweep_build") == 1
oracle_py.lib()
rng = np.random.default_rng(5)
ALPHA8 = np.frombuffer(b"ACGTacgt", dtype=np.uint8)


def setv(key, value):
    ctx.require_variant(key, value)


ok = True
# ---- against the oracle, through the device entry point ----
setv("scan_impl", 7)
COMBOS = [(sh, ps, U, pack) for sh in (5, 4, 3, 1, 2) for ps in (0, 1) for U in (2, 4) for pack in (0, 1, 2) if not (sh == 5 and pack == 2)] + [(0, 1, 2, 0), (0, 0, 2, 1)]
for sh, ps, U, pack in COMBOS:
    setv("scan_mfma_shift", sh)
    setv("scan_mfma_persist", ps)
    setv("scan_mfma_pack", pack)
    setv("scan_mfma_unroll", U)
    good = True
    for k in (1, 2, 15, 16, 17, 31, 32):
        for n in (k, k + 1, 1000, 1055, 1056, 1057, 2079, 2080, 2081, 3103, 3104, 4128, 4129, 5152, 5153, 8 * 1024 + 32, 12 * 1024 + 33, 65 * 1024 + 100, 200003, 3 * 10**6 + 77):
            if (pack != 1 or k not in (31, 32)) and n > 70000:
                continue
            s_ = ALPHA8[rng.integers(0, 8, size=n)]
            q = int(rng.integers(0, 1 << 62)) | (int(rng.integers(0, 4)) << 62)
            t = torch.from_numpy(s_).to(dev)
            d = torch.full((n + 16,), 0xEE, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            ctx.kmer_hdist_scan_dev(t, n, k, q, d)
            ctx.sync()
            want = oracle_py.kmer_hdist_scan(s_, k, q)
            got = d.cpu().numpy()
            same = np.array_equal(got[:n - k + 1], want) and bool((got[n - k + 1:] == 0xEE).all())
            if not same:
                bad = np.nonzero(got[:n - k + 1] != want)[0]
                print(f"MISMATCH shift {sh} persist {ps} U {U} pack {pack} k {k} n {n}: {bad.size} windows differ, first {bad[:5]}, got {got[bad[:5]]} want {want[bad[:5]]}", flush=True)
                good = ok = False
    print(f"shift {sh} persist {ps} U {U} pack {pack}: oracle parity {'ok' if good else 'FAILED'}", flush=True)
# first invalid byte
n = 50000
s = ALPHA8[rng.integers(0, 8, size=n)].copy()
for sh, ps, pos in [(sh, ps, pos) for sh in (5, 4, 3, 1, 2) for ps in (0, 1) for pos in (0, 15, 16, 1023, 1024, 1040, 1055, 1056, 4095, 4096, 4097, 30000, 49600, 49999)]:
    setv("scan_mfma_shift", sh)
    setv("scan_mfma_persist", ps)
    tt = s.copy()
    tt[pos] = ord("N")
    if pos + 9 < n:
        tt[pos + 9] = ord("X")
    try:
        ctx.kmer_hdist_scan(tt, 31, 0)
        print(f"invalid byte at {pos}: NOT reported")
        ok = False
    except bitnuc_amd.NucleotideError as e:
        if (e.byte, e.index) != (ord("N"), pos):
            print(f"invalid byte at {pos}: reported {(e.byte, e.index)}")
            ok = False
# fused count
for sh, U in ((5, 2), (5, 4), (4, 2), (4, 4), (3, 2), (3, 4), (1, 2), (1, 4), (2, 2), (2, 4), (0, 2)):
  setv("scan_mfma_shift", sh)
  setv("scan_mfma_unroll", U)
  setv("scan_mfma_count_persist", (sh + U) % 2)  # both grid forms of the fused count get their share of the sizes
  cnt = torch.zeros(1, dtype=torch.int64, device=dev)
  for n in (31, 1056, 2081, 4129, 16 * 1024 + 31, 100003, 1 << 20):
    s = ALPHA8[rng.integers(0, 8, size=n)]
    t = torch.from_numpy(s).to(dev)
    for k, tau in ((31, 20), (31, 0), (32, 24), (16, 9), (1, 0), (7, 7)):
        if n < k:
            continue
        q = int(rng.integers(0, 1 << 62)) & ((1 << (2 * k)) - 1)
        if tau == 0 and n > 2000:
            q = oracle_py.as_2bit(s[1500:1500 + k])
        torch.cuda.synchronize()
        ctx.kmer_hdist_count_dev(t, n, k, q, tau, cnt)
        ctx.sync()
        want = int((oracle_py.kmer_hdist_scan(s, k, q) <= tau).sum())
        if int(cnt.item()) != want:
            print(f"COUNT MISMATCH shift {sh} U {U} n {n} k {k} tau {tau}: {int(cnt.item())} != {want}")
            ok = False
print("small-size parity (oracle, invalid bytes, count):", "ok" if ok else "FAILED", flush=True)

# ---- full size ----
n, k = (10**8 if quick else 10**9), 31
q = 0x1B1B1B1B1B1B1B1B & ((1 << 62) - 1)
ref = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(ref, n, 0xB17C0DE)
dists = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
ctx.sync()
setv("scan_impl", 1)
base = torch.full((n,), 0xEE, dtype=torch.uint8, device=dev)
ctx.kmer_hdist_scan_dev(ref, n, k, q, base)
ctx.sync()
for sh, ps, U, pack in ((5, 0, 4, 1), (5, 1, 4, 0), (5, 0, 2, 1), (4, 0, 4, 1), (4, 1, 2, 0), (3, 0, 4, 1), (3, 1, 2, 0), (1, 0, 4, 1), (1, 1, 4, 0), (2, 0, 4, 1), (2, 1, 2, 2), (0, 1, 2, 1)):
    setv("scan_impl", 7)
    setv("scan_mfma_shift", sh)
    setv("scan_mfma_persist", ps)
    setv("scan_mfma_unroll", U)
    setv("scan_mfma_pack", pack)
    d = torch.full((n,), 0xEE, dtype=torch.uint8, device=dev)
    ctx.kmer_hdist_scan_dev(ref, n, k, q, d)
    ctx.sync()
    same = torch.equal(d, base)
    ok = ok and same
    print(f"{n} bases, k=31: mfma shift {sh} persist {ps} U {U} pack {pack} == bit-plane scan: {same}", flush=True)
    del d
for impl in (1, 7):
    setv("scan_impl", impl)
    ctx.kmer_hdist_count_dev(ref, n, k, q, 18, cnt)
    ctx.sync()
    print(f"count(d <= 18) impl {impl}: {int(cnt.item())}   (distance bytes: {int((base[:n - k + 1] <= 18).sum().item())})", flush=True)
del base

FORMS = [("bit-plane scan2 GEN1 (ships r04)", dict(scan_impl=1))]
CENTRE = dict(shift=4, persist=0, U=4, pack=1, pol=3, grid=4, cp=1)
seen = set()
for f in [CENTRE] + [dict(CENTRE, **d) for d in (dict(shift=5), dict(shift=5, pack=0), dict(shift=5, U=2), dict(shift=5, persist=1), dict(shift=5, cp=0), dict(shift=5, pack=0, cp=0), dict(shift=5, U=2, cp=0),
                                                   dict(cp=0), dict(U=2, cp=0), dict(shift=3), dict(shift=1), dict(pack=0), dict(persist=1))]:
    tup = tuple(sorted(f.items()))
    if tup in seen or (quick and f != CENTRE):
        continue
    seen.add(tup)
    FORMS.append((f"mfma shift{f['shift']} persist{f['persist']} U{f['U']} pack{f['pack']} countpersist{f['cp']}",
                  dict(scan_impl=7, scan_mfma_shift=f["shift"], scan_mfma_persist=f["persist"], scan_mfma_unroll=f["U"], scan_mfma_pack=f["pack"], scan_mfma_policy=f["pol"], scan_mfma_grid=f["grid"], scan_mfma_count_persist=f["cp"])))
FORMS.append(("mfma shift0 (global re-loads) persist1 U2 pack1", dict(scan_impl=7, scan_mfma_shift=0, scan_mfma_persist=1, scan_mfma_unroll=2, scan_mfma_pack=1, scan_mfma_policy=3, scan_mfma_grid=4)))
flip = [0]


def use(form):
    for key, v in form.items():
        setv(key, v)


def scan():
    flip[0] ^= 1
    ctx.kmer_hdist_scan_dev(ref, n, k, q, dists[flip[0]])


def count():
    ctx.kmer_hdist_count_dev(ref, n, k, q, 18, cnt)


def burst(fn, B=8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    a.record(stream)
    for _ in range(B):
        fn()
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / B


def queue(fn, N=96):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    ev[0].record(stream)
    for i in range(N):
        fn()
        ev[i + 1].record(stream)
    torch.cuda.synchronize()
    us = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(N)]
    return sum(us) / N, sum(us[-16:]) / 16, max(us[1:])


alg = 2 * (n - k + 1)
for what, fn, bytes_ in (("scan", scan, alg), ("count", count, alg // 2)):
    res = {name: [] for name, _ in FORMS}
    for rnd in range(7):
        for name, form in FORMS:
            use(form)
            t = burst(fn)
            if rnd:
                res[name].append(t)
    for name, _ in FORMS:
        m = statistics.median(res[name])
        print(f"{what:5s} bursts {name:52s} {m*1e3:7.1f} us  {bytes_/m/1e6:6.0f} GB/s  {bytes_/m/8e7:5.1f} % of 8 TB/s", flush=True)
    for rep in range(2):
        for name, form in FORMS[:1] + [f for f in FORMS[1:] if f[1]["scan_mfma_shift"] in (4, 5) and f[1]["scan_mfma_persist"] == 0]:
            use(form)
            torch.cuda.synchronize()
            time.sleep(1.0)  # from idle
            mean, settled, worst = queue(fn)
            print(f"{what:5s} queue{rep} {name:52s} mean of 96 {mean:6.1f} us ({bytes_/mean/8e4:4.1f} %)  last 16 {settled:6.1f} us ({bytes_/settled/8e4:4.1f} %)  slowest {worst:6.1f} us", flush=True)
setv("scan_impl", 1)
print("all outputs equal" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
