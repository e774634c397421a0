#!/usr/bin/env python3
"""bench.py's timed step (encode set r, decode set r+1, three rotating sets, cache-cold) with the nine buffers as nine separate allocations
(what torch.empty gives: one hipMalloc each) against the same nine buffers carved from ONE allocation, 2 MiB aligned.  Several fresh rounds of
each in one process, interleaved, so that the allocator's placement varies.  tools/ab_placement.py showed: inside one allocation the codec
does not care where its buffers sit (194-197 us), separate allocations vary by +-4 % (195-205 us)."""
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
nw = (n + 31) // 32
R = 3
A = 2 << 20


def up(x):
    return (x + A - 1) // A * A


def make_separate():
    seqs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(R)]
    words = [torch.empty(nw, dtype=torch.int64, device=dev) for _ in range(R)]
    backs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(R)]
    return (seqs, words, backs), [t.data_ptr() for t in seqs], [t.data_ptr() for t in words], [t.data_ptr() for t in backs]


def make_arena():
    total = R * (2 * up(n) + up(8 * nw)) + A
    arena = torch.empty(total, dtype=torch.uint8, device=dev)
    p = up(arena.data_ptr())
    seqs, words, backs = [], [], []
    for _ in range(R):
        seqs.append(p); p += up(n)
    for _ in range(R):
        words.append(p); p += up(8 * nw)
    for _ in range(R):
        backs.append(p); p += up(n)
    return arena, seqs, words, backs


def run(seqs, words, backs, steps=40):
    for r in range(R):
        ctx.nucgen_dev(seqs[r], n, 0xB17C0DE + r)
        ctx.encode_dev(seqs[r], n, words[r])
    ctx.sync()

    def step(i):
        r = i % R
        d = (r + 1) % R
        ctx.encode_dev(seqs[r], n, words[r])
        ctx.decode_dev(words[d], nw, n, backs[d])
    for i in range(6):
        step(i)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * steps + 1)]
    ev[0].record(stream)
    for i in range(steps):
        r = i % R
        d = (r + 1) % R
        ctx.encode_dev(seqs[r], n, words[r]); ev[2 * i + 1].record(stream)
        ctx.decode_dev(words[d], nw, n, backs[d]); ev[2 * i + 2].record(stream)
    torch.cuda.synchronize()
    enc = [ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(steps)]
    dec = [ev[2 * i + 1].elapsed_time(ev[2 * i + 2]) for i in range(steps)]
    return ev[0].elapsed_time(ev[2 * steps]) / steps, statistics.mean(enc), statistics.mean(dec), [statistics.mean(enc[r::R]) for r in range(R)]


hold = []
for rnd in range(6):
    keep, s, w, b = make_separate()
    t = run(s, w, b)
    print(f"round {rnd} separate allocations: step {t[0]:.4f} ms  encode {t[1]*1e3:.1f} us (per set {' '.join(f'{x*1e3:.1f}' for x in t[3])})  decode {t[2]*1e3:.1f} us", flush=True)
    del keep
    keep, s, w, b = make_arena()
    t = run(s, w, b)
    print(f"round {rnd} one arena           : step {t[0]:.4f} ms  encode {t[1]*1e3:.1f} us (per set {' '.join(f'{x*1e3:.1f}' for x in t[3])})  decode {t[2]*1e3:.1f} us", flush=True)
    del keep
    hold.append(torch.empty((rnd + 1) * 53 * (1 << 20), dtype=torch.uint8, device=dev))  # perturb the next round's placement
    torch.cuda.empty_cache()
ctx.close()
