#!/usr/bin/env python3
"""ONE BASELINE side config at full size and nothing else (for rocprofv3 --kernel-trace --stats / --pmc passes whose per-kernel averages must
not be mixed with small launches of the same kernel):
  cfg3   10^8 dense 31-mers as_2bit -> u64 (kmer_dense_kernel): 3.1 GB read + 0.8 GB written per launch = 39 B per k-mer
  cfg5   sliding 31-mer pack + Hamming distance to one query over 10^9 bases (kmer_scan_seg_mfma_kernel since round 5; kmer_scan2_kernel before): 1 B read + 1 B written per window
N launches in ONE queue (default 24 for cfg3, 96 for cfg5; the PMC passes use 12), two output buffers in rotation (the 256 MiB Infinity
Cache holds neither the input nor an output).  cfg5's kernel is VALU-issue bound and the chip lowers its clock under it for the
first ~40 launches of a queue (profiles/r04_launch_series.txt): a long queue makes the trace's AVERAGE the settled rate while its
first launches still show the transient; HBM traffic per launch does not depend on the clock.
The library must exist already (tools/prof_r04.sh builds it first): a process under the profiler does not start a compiler."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd
from bitnuc_amd import build

build.ensure_built(build=False)
which = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
launches = int(sys.argv[2]) if len(sys.argv) > 2 else (24 if which == "cfg3" else 96)  # cfg5 / cfg5count: a queue of 96 from an idle chip
dev = torch.device("cuda:0")
ctx = bitnuc_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream)
SEED, k = 0xB17C0DE, 31
if which == "cfg3":
    count = 10**8
    seq = torch.empty(count * k, dtype=torch.uint8, device=dev)
    ctx.nucgen_dev(seq, count * k, SEED + 100)
    outs = [torch.empty(count, dtype=torch.int64, device=dev) for _ in range(2)]
    ctx.sync()
    for i in range(launches):
        ctx.as_2bit_batch_dev(seq, k, k, count, outs[i & 1])
elif which == "cfg5":
    n = 10**9
    ref = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.nucgen_dev(ref, n, SEED)
    outs = [torch.empty(n - k + 1, dtype=torch.uint8, device=dev) for _ in range(2)]
    q = 0x1B1B1B1B1B1B1B1B & ((1 << 62) - 1)
    ctx.sync()
    for i in range(launches):
        ctx.kmer_hdist_scan_dev(ref, n, k, q, outs[i & 1])
elif which == "cfg5count":  # SURVEY 8d cfg 5's fused output: only the count of windows with d <= tau leaves the chip (1 B read per window)
    n = 10**9
    ref = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.nucgen_dev(ref, n, SEED)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    q = 0x1B1B1B1B1B1B1B1B & ((1 << 62) - 1)
    ctx.sync()
    for i in range(launches):
        ctx.kmer_hdist_count_dev(ref, n, k, q, 8, cnt)
else:
    raise SystemExit("usage: run_cfg35.py cfg3|cfg5|cfg5count [launches]")
ctx.sync()
ctx.close()
