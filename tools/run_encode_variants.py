#!/usr/bin/env python3
"""Bulk encode of 10^9 bases with a few variants of the evidence build, cache-cold rotation (for rocprofv3 --pmc runs):
39 = shipped (4-byte nt stores), 14 = round 1's (4-byte plain stores), 50 / 62 = quad transpose, 16-byte nt stores, 128 / 256
threads, 56 = quad transpose, 16-byte plain stores.  The kernel names carry the template arguments."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd
from bitnuc_amd import build as _build

dev = torch.device("cuda:0")
ctx = bitnuc_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream, lib_path=_build.ensure_built(sweep=True))
n = 10**9
seqs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(3)]
words = [torch.empty(n // 32, dtype=torch.int64, device=dev) for _ in range(3)]
for r in range(3):
    ctx.nucgen_dev(seqs[r], n, 7 + r)
ctx.sync()
for v in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "39,14,50,62,56").split(",")]:
    ctx.require_variant("encode", v)
    for i in range(6):
        ctx.encode_dev(seqs[i % 3], n, words[i % 3])
    ctx.sync()
