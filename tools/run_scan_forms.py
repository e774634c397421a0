#!/usr/bin/env python3
"""A few launches of each config-5 scan / fused-count form on 10^9 bases (evidence build), for rocprofv3 --pmc passes (tools/pmc_scan_mfma.sh):
the kernels are told apart by their names and template arguments in the counter CSV.  usage: run_scan_forms.py [launches=4]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd
from bitnuc_amd import build

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
ctx = bitnuc_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream, lib_path=build.ensure_built(sweep=True, build=False))
n, k = 10**9, 31
q = 0x1B1B1B1B1B1B1B1B & ((1 << 62) - 1)
ref = torch.empty(n, dtype=torch.uint8, device=dev)
dist = torch.empty(n, dtype=torch.uint8, device=dev)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
ctx.nucgen_dev(ref, n, 0xB17C0DE)
ctx.sync()
FORMS = [dict(scan_impl=1)] + [dict(scan_impl=7, scan_mfma_shift=sh, scan_mfma_persist=0, scan_mfma_unroll=4, scan_mfma_pack=1) for sh in (4, 3, 1)]
for f in FORMS:
    for key, v in f.items():
        ctx.require_variant(key, v)
    for _ in range(N):
        ctx.kmer_hdist_scan_dev(ref, n, k, q, dist)
    for _ in range(N):
        ctx.kmer_hdist_count_dev(ref, n, k, q, 18, cnt)
    ctx.sync()
ctx.require_variant("scan_mfma_count_form", 1)
ctx.require_variant("scan_mfma_count_rounds", 3)
ctx.require_variant("scan_mfma_count_grid", 18)
for emit in (0, 1, 2):  # the four-channel count: how a round's 1024 distances become a count (kmer_count_mfma_kernel<3, true, EMIT>)
    ctx.require_variant("scan_mfma_count_emit", emit)
    for _ in range(N):
        ctx.kmer_hdist_count_dev(ref, n, k, q, 18, cnt)
    ctx.sync()
ctx.require_variant("scan_mfma_count_form", 2)  # three channels per base (kmer_count3_mfma_kernel<4, true>, ships)
ctx.require_variant("scan_mfma_count_rounds", 4)
ctx.require_variant("scan_mfma_count_grid", 12)
for _ in range(N):
    ctx.kmer_hdist_count_dev(ref, n, k, q, 18, cnt)
ctx.sync()
ctx.require_variant("scan_impl", 8)  # the scan in the segment tiling (kmer_scan_seg_mfma_kernel<3, 4>, ships)
for _ in range(N):
    ctx.kmer_hdist_scan_dev(ref, n, k, q, dist)
ctx.sync()
print("done", int(cnt.item()))
