#!/usr/bin/env python3
"""Every 31-base window of 10^9 bases (for rocprofv3 runs): the strip kernel of rounds 1-2 (rounds of 992 windows through a
wave-private LDS strip) and the line-aligned kernel without LDS, two alternating outputs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd
from bitnuc_amd import build as _build

dev = torch.device("cuda:0")
ctx = bitnuc_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream, lib_path=_build.ensure_built(sweep=True))
n, k = 10**9, 31
seq = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(seq, n, 3)
outs = [torch.empty(n - k + 1, dtype=torch.int64, device=dev) for _ in range(2)]
ctx.sync()
for impl, u in ((0, 1), (1, 4)):
    ctx.require_variant("slide_impl", impl)
    ctx.require_variant("slide_rounds", 1)
    ctx.require_variant("slide2_rounds", u)
    for i in range(6):
        ctx.as_2bit_batch_dev(seq, k, 1, n - k + 1, outs[i & 1])
    ctx.sync()
