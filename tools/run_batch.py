#!/usr/bin/env python3
"""Run the ragged-batch kernels on 150-base reads a few times (for rocprofv3 --kernel-trace / --pmc runs): layout plan
build, plan encode / decode, and the table-driven encode / decode."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd

dev = torch.device("cuda:0")
ctx = bitnuc_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 150
count = 10**9 // L
n = L * count
seq = torch.empty(n, dtype=torch.uint8, device=dev)
back = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(seq, n, 1)
off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L
wo = torch.empty(count + 1, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
total = ctx.batch_word_offsets_dev(off, count, wo)
words = torch.empty(total, dtype=torch.int64, device=dev)
plan = bitnuc_amd.BatchPlan(ctx, off, count)
for _ in range(8):
    plan.encode_dev(seq, words)
    plan.decode_dev(words, back)
    ctx.encode_batch_dev(seq, off, wo, count, total, words)
    ctx.decode_batch_dev(words, wo, off, count, total, back)
ctx.sync()
assert torch.equal(seq, back)
