#!/usr/bin/env python3
"""Run encode_fixed / decode_fixed on 150-base reads a few times (for rocprofv3 runs)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd

dev = torch.device("cuda:0")
ctx = bitnuc_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 150
count = 10**9 // L
seq = torch.empty(L * count, dtype=torch.uint8, device=dev)
back = torch.empty(L * count, dtype=torch.uint8, device=dev)
words = torch.empty(count * ((L + 31) // 32), dtype=torch.int64, device=dev)
ctx.nucgen_dev(seq, L * count, 1)
for _ in range(6):
    ctx.encode_fixed_dev(seq, L, L, count, words)
    ctx.decode_fixed_dev(words, L, L, count, back)
ctx.sync()
assert torch.equal(seq, back)
