#!/usr/bin/env python3
"""Does the PLACEMENT of the buffers move the bulk codec?  The rocprofv3 trace of the timed step shows a period-3 pattern in the encode's
per-launch time (193 .. 214 us: the three rotating buffer sets), i.e. +-4 % by where the allocator happened to put a set.  Here one arena
is carved by hand: input at a fixed place, output at input_end + delta for a ladder of deltas (and the whole pair shifted by a ladder of
bases), sustained bursts over two pairs of equal geometry.  usage: ab_placement.py"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream)
n = 10**9
nw = n // 32
GB = 1 << 30
arena = torch.empty(8 * GB, dtype=torch.uint8, device=dev)
base = arena.data_ptr()
base += (-base) % (2 << 20)  # 2 MiB aligned start


def once(fn, B=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(0)
    a.record(stream)
    for i in range(B):
        fn(i + 1)
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / B


def geometry(shift, delta):
    """two pairs (seq, words[, back]) of the same relative geometry, 3 GiB apart"""
    pairs = []
    for p in range(2):
        s = base + p * 3 * GB + shift
        w = s + n + ((-n) % 4096) + delta
        pairs.append((s, w))
    return pairs


print("encode: input at arena + shift, words at input_end (4 KiB aligned) + delta")
for shift in (0, 4096, 1 << 16, 1 << 20):
    row = []
    for delta in (0, 4096, 8192, 1 << 16, 1 << 18, 1 << 20, (1 << 20) + 4096, 2 << 20, (2 << 20) + (1 << 16), 16 << 20, (16 << 20) + 12288):
        pairs = geometry(shift, delta)
        for s, w in pairs:
            ctx.nucgen_dev(s, n, 0xB17C0DE)
        ctx.sync()
        ts = []
        for _ in range(5):
            ts.append(once(lambda i: ctx.encode_dev(pairs[i & 1][0], n, pairs[i & 1][1])))
        row.append((delta, statistics.median(ts[1:])))
    print(f"shift {shift:>8d}: " + "  ".join(f"+{d>>10}K {t*1e3:.1f}us" for d, t in row), flush=True)
print("decode: words at arena + shift, output at words_end (4 KiB aligned) + delta")
for shift in (0, 1 << 16):
    row = []
    for delta in (0, 4096, 1 << 16, 1 << 20, 2 << 20, (2 << 20) + (1 << 16), (16 << 20) + 12288):
        pairs = []
        for p in range(2):
            w = base + p * 3 * GB + shift
            o = w + 8 * nw + ((-8 * nw) % 4096) + delta
            pairs.append((w, o))
        tmp = torch.empty(n, dtype=torch.uint8, device=dev)
        ctx.nucgen_dev(tmp, n, 0xB17C0DE)
        for w, o in pairs:
            ctx.encode_dev(tmp, n, w)
        ctx.sync()
        del tmp
        ts = []
        for _ in range(5):
            ts.append(once(lambda i: ctx.decode_dev(pairs[i & 1][0], nw, n, pairs[i & 1][1])))
        row.append((delta, statistics.median(ts[1:])))
    print(f"shift {shift:>8d}: " + "  ".join(f"+{d>>10}K {t*1e3:.1f}us" for d, t in row), flush=True)
# the same geometry in fresh allocations, five times: what the allocator's placement does
print("fresh torch allocations (seq, words), encode:")
for rep in range(6):
    seqs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
    words = [torch.empty(nw, dtype=torch.int64, device=dev) for _ in range(2)]
    for s in seqs:
        ctx.nucgen_dev(s, n, 0xB17C0DE)
    ctx.sync()
    ts = [once(lambda i: ctx.encode_dev(seqs[i & 1], n, words[i & 1])) for _ in range(4)]
    print(f"  rep {rep}: {statistics.median(ts[1:])*1e3:.1f} us   seq % 2MiB = {[s.data_ptr() % (2<<20) for s in seqs]}  words % 2MiB = {[w.data_ptr() % (2<<20) for w in words]}", flush=True)
    keep = torch.empty((rep + 1) * 37 * (1 << 20), dtype=torch.uint8, device=dev)  # perturb the next round's placement
    del seqs, words
ctx.close()
