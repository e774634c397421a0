#!/usr/bin/env python3
"""Pipelined host-pointer path, round 3: what the PCIe link gives when H2D and D2H run at once (encode moves 1 B in and 0.25 B
out per base, decode the reverse: is the "pinned rate" of ONE direction the right denominator?), how the staging pools were
sized (bitnuc_host_pipe_info), and encode / decode of 10^9 bases under a few forced settings."""
import os

os.environ.setdefault("BITNUC_PIPE_IMPL", "staged")  # this tool studies the STAGED engine's thread budget / placement (the direct engine ships: tools/ab_pipe_impl.py)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bitnuc_amd

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
CH = 32 << 20
pin_a = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
pin_b = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
dev_a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
dev_b = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def duplex(in_bytes, out_bytes, chunks=30):
    """`chunks` x (H2D of in_bytes on one stream, D2H of out_bytes on another), all queued at once; GB/s of each side."""
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for c in range(chunks):
            if in_bytes:
                with torch.cuda.stream(s1):
                    dev_a[c * in_bytes:(c + 1) * in_bytes].copy_(pin_a[c * in_bytes:(c + 1) * in_bytes], non_blocking=True)
            if out_bytes:
                with torch.cuda.stream(s2):
                    pin_b[c * out_bytes:(c + 1) * out_bytes].copy_(dev_b[c * out_bytes:(c + 1) * out_bytes], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        best = dt if best is None or dt < best else best
    return chunks * in_bytes / best / 1e9, chunks * out_bytes / best / 1e9, best / chunks * 1e3


for label, a, b in (("H2D alone, 32 MiB chunks", CH, 0), ("D2H alone, 32 MiB chunks", 0, CH), ("encode's mix: 32 MiB in + 8 MiB out", CH, CH // 4),
                    ("decode's mix: 8 MiB in + 32 MiB out", CH // 4, CH), ("both ways, 32 MiB + 32 MiB", CH, CH)):
    i, o, ms = duplex(a, b)
    print(f"{label:42s} H2D {i:5.1f} GB/s   D2H {o:5.1f} GB/s   {ms:.3f} ms per chunk pair", flush=True)
del pin_a, pin_b, dev_a, dev_b
seq = np.repeat(np.frombuffer(b"ACGT", dtype=np.uint8)[np.frombuffer(np.random.default_rng(1).bytes(n // 4 + 1), dtype=np.uint8) & 3], 4)[:n].copy()
w = np.zeros((n + 31) // 32, dtype=np.uint64)  # caller-owned, already touched
back = np.zeros(n, dtype=np.uint8)
print(f"cores visible: {len(os.sched_getaffinity(0))}")


def run(label, env, null_stream=False):
    for k in ("BITNUC_HOST_THREADS", "BITNUC_HOST_THREADS_LIGHT", "BITNUC_PIPE_CHUNK_MB", "BITNUC_PIPE_CALIBRATE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    # null_stream: the context rides on torch's current stream = the legacy NULL stream (what bench.py's main context does)
    ctx = bitnuc_amd.Context(0, stream=torch.cuda.current_stream().cuda_stream) if null_stream else bitnuc_amd.Context(0)
    info = ctx.host_pipe_info()
    te, td = [], []
    for _ in range(4):
        t = time.perf_counter()
        ctx.encode_into(seq, w)
        te.append(time.perf_counter() - t)
        t = time.perf_counter()
        ctx.decode_into(w, n, back)
        td.append(time.perf_counter() - t)
    ok = bool(np.array_equal(back, seq))
    th = f"enc {info['encode_stage_in_threads']}+{info['encode_hand_back_threads']} dec {info['decode_stage_in_threads']}+{info['decode_hand_back_threads']}"
    print(f"{label:40s} [{th:20s}] encode {n / min(te[1:]) / 1e9:5.1f}   decode {n / min(td[1:]) / 1e9:5.1f} Gbases/s   {'ok' if ok else 'MISMATCH'}", flush=True)
    if not env:
        print("   pipe:", info, flush=True)
    ctx.close()


run("default", {})
run("default, context on the NULL stream", {}, null_stream=True)
run("default again", {})
run("NULL stream again", {}, null_stream=True)
run("heavy 8, light 2", {"BITNUC_HOST_THREADS": "8", "BITNUC_HOST_THREADS_LIGHT": "2"})
run("heavy 8, light 4", {"BITNUC_HOST_THREADS": "8", "BITNUC_HOST_THREADS_LIGHT": "4"})
run("heavy 11, light 4", {"BITNUC_HOST_THREADS": "11", "BITNUC_HOST_THREADS_LIGHT": "4"})
run("heavy 6, light 2", {"BITNUC_HOST_THREADS": "6", "BITNUC_HOST_THREADS_LIGHT": "2"})
run("heavy 8, light 2, 64 Mi chunks", {"BITNUC_HOST_THREADS": "8", "BITNUC_HOST_THREADS_LIGHT": "2", "BITNUC_PIPE_CHUNK_MB": "64"})
run("heavy 8, light 2, 16 Mi chunks", {"BITNUC_HOST_THREADS": "8", "BITNUC_HOST_THREADS_LIGHT": "2", "BITNUC_PIPE_CHUNK_MB": "16"})
