#!/usr/bin/env python3
"""Do two streams that advance in lockstep at the SAME offset of two equally aligned buffers collide in the HBM channel / bank
hash?  hdist_dev(a, b) and the copy probe with the second buffer shifted by 0 .. a few MiB inside its allocation."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bitnuc_amd

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream)
n = 10**9
nw = n // 32
SL = 8 << 20
seq = torch.empty(n, dtype=torch.uint8, device=dev)
bufs = [torch.empty(nw * 8 + SL, dtype=torch.uint8, device=dev) for _ in range(4)]
res1 = torch.zeros(1, dtype=torch.int32, device=dev)
for r, b in enumerate(bufs):
    ctx.nucgen_dev(seq, n, 21 + r)
    ctx.encode_dev(seq, n, b.data_ptr())
ctx.sync()
big = [torch.empty((1 << 30) + SL, dtype=torch.uint8, device=dev) for _ in range(2)]
print("shift of the second stream   hdist us (GB/s)      copy 1 GiB us (GB/s read+written)")
for shift in (0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, (2 << 20) + 8192 + 256, 3 * 1024 * 1024 + 12288):
    # second operands re-encoded at the shifted address so that a and b stay valid packed words
    for r in (1, 3):
        ctx.nucgen_dev(seq, n, 21 + r)
        ctx.encode_dev(seq, n, bufs[r].data_ptr() + shift)
    ctx.sync()
    th, tc = [], []
    for rnd in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ctx.hdist_dev(bufs[0].data_ptr(), nw, bufs[1].data_ptr() + shift, nw, n, res1)
        a.record(stream)
        for i in range(8):
            ctx.hdist_dev(bufs[(i & 1) * 2].data_ptr(), nw, bufs[(i & 1) * 2 + 1].data_ptr() + shift, nw, n, res1)
        b.record(stream)
        torch.cuda.synchronize()
        if rnd:
            th.append(a.elapsed_time(b) / 8)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for i in range(8):
            ctx.stream_probe_dev(1 | 8 | 16, big[0].data_ptr(), big[1].data_ptr() + shift, 1 << 30)
        b.record(stream)
        torch.cuda.synchronize()
        if rnd:
            tc.append(a.elapsed_time(b) / 8)
    mh, mc = statistics.median(th), statistics.median(tc)
    print(f"{shift:10d} B   {mh * 1e3:7.1f} ({16 * nw / mh / 1e6:5.0f})   {mc * 1e3:7.1f} ({2 * (1 << 30) / mc / 1e6:5.0f})")
