#!/usr/bin/env python3
"""Would the step gain from issuing its two kernels on TWO streams (encode of set r and decode of another set are independent)?
One stream, as bench.py does it, against two streams with the cross-stream dependencies the rotation needs (4 sets: an encode waits
for the decode two steps back that read its output buffer, a decode for the encode two steps back that wrote its input)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bitnuc_amd

dev = torch.device("cuda:0")
n = 10**9
nw = n // 32
R = 4
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
cA = bitnuc_amd.Context(0, stream=sA.cuda_stream)
cB = bitnuc_amd.Context(0, stream=sB.cuda_stream)
for c in (cA, cB):
    c.require_variant("force_gpu", 1)
seqs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(R)]
words = [torch.empty(nw, dtype=torch.int64, device=dev) for _ in range(R)]
backs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(R)]
for r in range(R):
    cA.nucgen_dev(seqs[r], n, 5 + r)
    cA.encode_dev(seqs[r], n, words[r])
cA.sync()
STEPS = 100


def one_stream():
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(sA)
    for i in range(STEPS):
        cA.encode_dev(seqs[i % R], n, words[i % R])
        cA.decode_dev(words[(i + 2) % R], nw, n, backs[(i + 2) % R])
    t1.record(sA)
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / STEPS


def two_streams():
    eE = [torch.cuda.Event() for _ in range(STEPS)]
    eD = [torch.cuda.Event() for _ in range(STEPS)]
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(sA)
    sB.wait_event(t0)
    for i in range(STEPS):
        if i >= 2:
            sA.wait_event(eD[i - 2])  # the decode that read words[i % R]
        cA.encode_dev(seqs[i % R], n, words[i % R])
        eE[i].record(sA)
        if i >= 2:
            sB.wait_event(eE[i - 2])  # the encode that wrote words[(i + 2) % R]
        cB.decode_dev(words[(i + 2) % R], nw, n, backs[(i + 2) % R])
        eD[i].record(sB)
    sA.wait_event(eD[STEPS - 1])
    t1.record(sA)
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / STEPS


res = {"one": [], "two": []}
for rnd in range(5):
    res["one"].append(one_stream())
    res["two"].append(two_streams())
cA.sync(); cB.sync()
ok = all(torch.equal(seqs[r], backs[r]) for r in range(R))
for k, v in res.items():
    m = statistics.median(v[1:])
    print(f"{k} stream(s): {m:.4f} ms per step  {2 * n / m / 1e6:7.1f} Gbases/s  (runs: {[round(x, 4) for x in v]})")
print("round trips", "ok" if ok else "MISMATCH")
