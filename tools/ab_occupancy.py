#!/usr/bin/env python3
"""Does the bulk codec's plateau move with OCCUPANCY alone?  The shipped pair (encode 39: 128-thread workgroups, decode 22: 256)
with 0 .. 64 KiB of unused dynamic LDS per workgroup (evidence build knob dyn_lds): a CU then holds 160 KiB / dyn_lds workgroups
instead of 16 / 8, i.e. the moving front of the streams narrows and fewer bytes are in flight, nothing else changes.  The bench's
sustained rotation (decode reads words written two steps earlier), per-kernel HIP events."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd
from bitnuc_amd import build as _b

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream, lib_path=_b.ensure_built(sweep=True))
ENC = int(sys.argv[1]) if len(sys.argv) > 1 else 39   # evidence-build encode variant (39 = shipped; 15 / 32 = 8 groups in flight per lane)
DEC = int(sys.argv[2]) if len(sys.argv) > 2 else 22
ctx.require_variant("encode", ENC)
ctx.require_variant("decode", DEC)
n = 10**9
nw = n // 32
R = 3
seqs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(R)]
words = [torch.empty(nw, dtype=torch.int64, device=dev) for _ in range(R)]
backs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(R)]
for r in range(R):
    ctx.nucgen_dev(seqs[r], n, 0xB17C0DE + r)
    ctx.encode_dev(seqs[r], n, words[r])
ctx.sync()
res = {}
SET = [0, 5 * 1024, 10 * 1024, 16 * 1024, 20 * 1024, 32 * 1024, 40 * 1024, 64 * 1024]
for rnd in range(4):
    for lds in SET:
        ctx.require_variant("dyn_lds", lds)
        evs = []
        for i in range(12):
            r = i % R
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record(stream)
            ctx.encode_dev(seqs[r], n, words[r])
            e[1].record(stream)
            ctx.decode_dev(words[(r + 1) % R], nw, n, backs[(r + 1) % R])
            e[2].record(stream)
            evs.append(e)
        torch.cuda.synchronize()
        if rnd:
            res.setdefault(lds, []).append((evs[3][0].elapsed_time(evs[-1][2]) / (len(evs) - 3),
                                            statistics.mean(e[0].elapsed_time(e[1]) for e in evs[3:]), statistics.mean(e[1].elapsed_time(e[2]) for e in evs[3:])))
ctx.require_variant("dyn_lds", 0)
ctx.sync()
assert torch.equal(seqs[0], backs[0])
print(f"encode variant {ENC}, decode variant {DEC}")
print("dyn LDS per workgroup -> workgroups per CU at 128 / 256 threads   step ms   encode ms (GB/s)   decode ms (GB/s)")
for lds in SET:
    v = res[lds]
    tot, enc, dec = (statistics.median(x[k] for x in v) for k in range(3))
    we = min(16, (160 * 1024) // lds) if lds else 16
    wd = min(8, (160 * 1024) // lds) if lds else 8
    print(f"{lds:6d} B   {we:2d} / {wd:2d}   {tot:.4f}   {enc:.4f} ({1.25 * n / enc / 1e6:5.0f})   {dec:.4f} ({1.25 * n / dec / 1e6:5.0f})")
