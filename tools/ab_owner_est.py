#!/usr/bin/env python3
"""block_owner_kernel's first guess (owner_est 0 / 1 / 2), timed through encode_batch_dev calls under rocprofv3
--kernel-trace --stats: run as `rocprofv3 --kernel-trace --stats -d out -o t -- python3 tools/ab_owner_est.py MODE`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd
from bitnuc_amd import build

mode = int(sys.argv[1])
dev = torch.device("cuda:0")
ctx = bitnuc_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream, lib_path=build.ensure_built(sweep=True))  # the selectors below exist in the evidence build only
ctx.require_variant("owner_est", mode)
N = 10**9
seq = torch.empty(N, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(seq, N, 1)
for L in (150, 1000, 37):
    count = N // L
    off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L
    wo = torch.empty(count + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    total = ctx.batch_word_offsets_dev(off, count, wo)
    words = torch.empty(total, dtype=torch.int64, device=dev)
    for _ in range(10):
        ctx.encode_batch_dev(seq, off, wo, count, total, words)
    ctx.sync()
