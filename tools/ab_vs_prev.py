#!/usr/bin/env python3
"""In-process A/B of this build against another build of the library (normally the previous commit's), same box, interleaved
rounds, sustained bursts: plan encode / decode, fixed-length encode / decode and the bulk codec on L-base reads.
Build the other library first, e.g.
  git archive HEAD~1 bitnuc_amd/csrc include | tar -x -C /tmp/prev
  (cd /tmp/prev/bitnuc_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc \
        -o $REPO/bitnuc_amd/libbitnuc_hip_prev.so runtime.hip codec.hip kmer.hip batch.hip analysis.hip comm.hip -ldl -lpthread)
  (commits before round 3 have the single unit bitnuc_hip.hip instead)
usage: ab_vs_prev.py [--prev PATH] [L ...]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bitnuc_amd

args = sys.argv[1:]
prev_path = os.path.join(ROOT, "bitnuc_amd", "libbitnuc_hip_prev.so")
if args and args[0] == "--prev":
    prev_path = args[1]
    args = args[2:]
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
ctxs = {"this": bitnuc_amd.Context(0, stream=stream.cuda_stream), "prev": bitnuc_amd.Context(0, stream=stream.cuda_stream, lib_path=prev_path)}
N = 10**9
seq = torch.empty(N, dtype=torch.uint8, device=dev)
ctxs["this"].nucgen_dev(seq, N, 0xB17C0DE)
BURST = 10


def once(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    a.record(stream)
    for _ in range(BURST):
        fn()
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / BURST


for L in [int(a) for a in args] or [150]:
    count = N // L
    wpr = (L + 31) // 32
    off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L
    torch.cuda.synchronize()
    plans = {k: bitnuc_amd.BatchPlan(c, off, count) for k, c in ctxs.items()}
    total = plans["this"].total_words
    words = [torch.empty(total + 64, dtype=torch.int64, device=dev) for _ in range(2)]
    backs = [torch.empty(N + 4096, dtype=torch.uint8, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    flip = [0]

    def alt(pair):
        flip[0] ^= 1
        return pair[flip[0]]

    ops = {
        "plan encode": lambda k: plans[k].encode_dev(seq, alt(words)),
        "plan decode": lambda k: plans[k].decode_dev(alt(words), alt(backs)),
        "fixed encode": lambda k: ctxs[k].encode_fixed_dev(seq, L, L, count, alt(words)),
        "fixed decode": lambda k: ctxs[k].decode_fixed_dev(alt(words), L, L, count, alt(backs)),
    }
    plans["this"].encode_dev(seq, words[0])
    plans["this"].encode_dev(seq, words[1])
    ctxs["this"].sync()
    res = {(o, k): [] for o in ops for k in ctxs}
    for rnd in range(7):
        for o, fn in ops.items():
            for k in ctxs:
                t = once(lambda: fn(k))
                if rnd >= 2:
                    res[(o, k)].append(t)
    # same results from both builds
    same = {}
    for o in ops:
        outs = []
        for k in ctxs:
            tgt = words if "encode" in o else backs
            flip[0] = 1  # alt() -> index 0
            if "decode" in o:
                backs[0].zero_()
                torch.cuda.synchronize()
                flip[0] = 1
                (plans[k].decode_dev(words[1], backs[0]) if o.startswith("plan") else ctxs[k].decode_fixed_dev(words[1], L, L, count, backs[0]))
                ctxs[k].sync()
                outs.append(backs[0][: L * count].clone())
            else:
                words[0].zero_()
                torch.cuda.synchronize()
                (plans[k].encode_dev(seq, words[0]) if o.startswith("plan") else ctxs[k].encode_fixed_dev(seq, L, L, count, words[0]))
                ctxs[k].sync()
                outs.append(words[0][:total].clone())
                plans["this"].encode_dev(seq, words[0])
                ctxs["this"].sync()
        same[o] = bool(torch.equal(outs[0], outs[1]))
    alg = L * count + 8 * total
    print(f"L={L}: {count} reads, {alg/1e9:.4f} GB algorithmic per launch")
    for o in ops:
        a, b = statistics.median(res[(o, "this")]), statistics.median(res[(o, "prev")])
        print(f"  {o:13s} this {a:.4f} ms {alg/a/1e6:6.0f} GB/s | prev {b:.4f} ms {alg/b/1e6:6.0f} GB/s | {100*(b/a-1):+5.1f} %  {'same output' if same[o] else 'DIFFERENT OUTPUT'}", flush=True)
    for p in plans.values():
        p.close()
    del plans, words, backs, off
