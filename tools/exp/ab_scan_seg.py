#!/usr/bin/env python3
"""The matrix-core scan (distance bytes): the shipped natural-layout tiling (scan_impl 7: six MFMAs per 1024 windows, half of each multiplying zeros, results
already in store order) against the count's tiling (scan_impl 8: segments of 32 windows x 32 shifts, four MFMAs, two v_permlane32_swap put the results in
store order), workgroups of 64 (ships) / 128 / 256 threads, trips of 4 / 3 rounds, and the count's tiling with three channels per base (scan_mfma_ch3 1: three MFMAs).  Why: profiles/r05_ablate_count_parts.txt -- the matrix pipe's power is what lowers the clock in a queue from idle.
Bursts of 8 and a 96-launch queue from an idle chip in groups of 8, interleaved three times; outputs compared; invalid bytes planted at round / trip
boundaries must be reported by every form."""
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
FORMS = [(8, 4, 0, 64), (8, 4, 0, 128), (8, 4, 0, 256), (8, 3, 0, 256), (8, 4, 1, 256), (7, 4, 0, 256)]  # (scan_impl, rounds per trip, 1 = three channels per base, threads per workgroup)  # (scan_impl, rounds per trip, 1 = three channels per base: kmer_scan_seg3_mfma_kernel)


def chan(f):
    return ("three channels" if f[2] else "four channels ") + (f", workgroups of {f[3]} threads" if len(f) > 3 else "")


def use(f):
    ctx.require_variant("scan_mfma_persist", 0)
    ctx.require_variant("scan_mfma_shift", 4)
    ctx.require_variant("scan_impl", f[0])
    ctx.require_variant("scan_mfma_unroll", f[1])
    ctx.require_variant("scan_mfma_ch3", f[2])
    ctx.require_variant("scan_mfma_block", f[3] if len(f) > 3 else 256)


base = None
for U in FORMS:
    use(U)
    d = torch.full((n,), 0xEE, dtype=torch.uint8, device=dev)
    ctx.kmer_hdist_scan_dev(ref, n, k, q, d)
    ctx.sync()
    if base is None:
        base = d
    else:
        print(f"form {U} == the shipped form at 10^9 bases: {torch.equal(d, base)}", flush=True)
import numpy as np
host = ref[:50000].cpu().numpy()
for f in FORMS:
    use(f)
    for pos in (0, 15, 16, 1023, 1024, 1040, 1055, 1056, 4095, 4096, 4097, 4127, 4128, 30000, 49999):
        b = host.copy()
        b[pos] = ord("N")
        if pos + 9 < b.size:
            b[pos + 9] = ord("X")
        tb = torch.from_numpy(b).to(dev)
        torch.cuda.synchronize()
        ctx.kmer_hdist_scan_dev(tb, b.size, 31, q, outs[0])
        try:
            ctx.sync()
            print(f"form {f}: invalid byte at {pos} NOT reported")
        except bitnuc_amd.NucleotideError as e:
            if (e.byte, e.index) != (ord("N"), pos):
                print(f"form {f}: invalid byte at {pos} reported as {(e.byte, e.index)}")
print("invalid bytes: checked", flush=True)
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
for rep in range(3):
    for U in FORMS:
        use(U)
        g = queue()
        m = statistics.median(res[U])
        print(f"scan_impl {U[0]} trips of {U[1]} {chan(U)}: bursts {m*1e3:6.1f} us ({alg/m/8e7:4.1f} %)   from idle: mean {sum(g)/len(g):6.1f} us ({alg/(sum(g)/len(g))/8e4:4.1f} %), settled {sum(g[-2:])/2:6.1f}, slowest group {max(g):6.1f}   groups: {' '.join(f'{x:.0f}' for x in g)}", flush=True)
