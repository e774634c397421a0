#!/usr/bin/env python3
"""The matrix-core scan (distance bytes): one trip per wave (the dispatcher walks 61 K workgroups) against a bounded grid whose waves walk several trips with register prefetch (scan_mfma_persist 1, scan_mfma_grid workgroups per CU): the strip
lets 4 / 6 / 9 workgroups share a CU.  Bursts of 8 and a 96-launch queue from an idle chip in groups of 8; outputs compared."""
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
outs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
ctx.sync()
FORMS = [(0, 4)] + [(1, g) for g in (32, 48, 64, 96, 128)]  # (persist, workgroups per CU)


def use(f):
    ctx.require_variant("scan_mfma_persist", f[0])
    ctx.require_variant("scan_mfma_grid", f[1])


base = None
for U in FORMS:
    use(U)
    d = torch.full((n,), 0xEE, dtype=torch.uint8, device=dev)
    ctx.kmer_hdist_scan_dev(ref, n, k, q, d)
    ctx.sync()
    if base is None:
        base = d
    else:
        print(f"form {U} == one trip per wave: {torch.equal(d, base)}", flush=True)
flip = [0]


def scan():
    flip[0] ^= 1
    ctx.kmer_hdist_scan_dev(ref, n, k, q, outs[flip[0]])


def burst(B=8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    scan()
    a.record(stream)
    for _ in range(B):
        scan()
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / B


def queue(N=96, every=8):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N // every + 1)]
    torch.cuda.synchronize()
    time.sleep(1.0)
    ev[0].record(stream)
    for i in range(N):
        scan()
        if (i + 1) % every == 0:
            ev[(i + 1) // every].record(stream)
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) * 1e3 / every for i in range(N // every)]


alg = 2 * (n - k + 1)
res = {U: [] for U in FORMS}
for rnd in range(7):
    for U in FORMS:
        use(U)
        t = burst()
        if rnd:
            res[U].append(t)
for rep in range(2):
    for U in FORMS:
        use(U)
        g = queue()
        m = statistics.median(res[U])
        print(f"{'one trip per wave      ' if not U[0] else 'bounded grid, ' + str(U[1]).rjust(2) + ' per CU'}: bursts {m*1e3:6.1f} us ({alg/m/8e7:4.1f} %)   from idle: mean {sum(g)/len(g):6.1f} us ({alg/(sum(g)/len(g))/8e4:4.1f} %), settled {sum(g[-2:])/2:6.1f}, slowest group {max(g):6.1f}   groups: {' '.join(f'{x:.0f}' for x in g)}", flush=True)
