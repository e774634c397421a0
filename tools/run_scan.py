#!/usr/bin/env python3
"""Run the config-5 scan and config-3 dense batch a few times (for rocprofv3 --pmc runs)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd

dev = torch.device("cuda:0")
ctx = bitnuc_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
n, k = 10**9, 31
ref = torch.empty(n, dtype=torch.uint8, device=dev)
dist = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.nucgen_dev(ref, n, 1)
kout = torch.empty(10**8 // 4, dtype=torch.int64, device=dev)
for _ in range(6):
    ctx.kmer_hdist_scan_dev(ref, n, k, 0x1B1B1B1B1B1B1B1B & ((1 << 62) - 1), dist)
    ctx.as_2bit_batch_dev(ref, k, k, 10**8 // 4, kout)
ctx.sync()
