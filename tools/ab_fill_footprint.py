#!/usr/bin/env python3
"""Streaming rate vs footprint: nt fill / read / copy probes over 0.25 .. 8 GB buffers, two alternating buffers, bursts sized to
move 32 GB each (so that what is still dirty in the 256 MiB Infinity Cache when the burst's last kernel ends is < 1 % of it)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream)
big = [torch.empty(8 << 30, dtype=torch.uint8, device=dev) for _ in range(2)]
for b in big:
    b.zero_()
torch.cuda.synchronize()
print("footprint   fill nt GB/s   fill plain GB/s   read nt GB/s   copy nt GB/s (read + written)")
for gb in (0.25, 1, 2, 4, 8):
    nbytes = int(gb * (1 << 30))
    burst = max(4, int(32 / gb))
    row = []
    for mode, moved in ((2 | 16, 1), (2, 1), (0 | 8, 1), (1 | 8 | 16, 2)):
        ts = []
        for rnd in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ctx.stream_probe_dev(mode, big[0], big[1], nbytes)
            a.record(stream)
            for i in range(burst):
                src, dst = (big[0], big[1]) if (mode & 7) == 1 else (big[i & 1], big[i & 1])
                ctx.stream_probe_dev(mode, src[(i & 1) * 0:], dst, nbytes)
            b.record(stream)
            torch.cuda.synchronize()
            if rnd:
                ts.append(a.elapsed_time(b) / burst)
        row.append(moved * nbytes / statistics.median(ts) / 1e6)
    print(f"{gb:5.2f} GB   {row[0]:7.0f}   {row[1]:7.0f}   {row[2]:7.0f}   {row[3]:7.0f}")
