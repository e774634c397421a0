#!/usr/bin/env python3
"""Host-pointer bulk path: where the time goes.  Sweeps the staging pool size and the chunk size of the pipelined
bitnuc_encode / bitnuc_decode (fresh context per setting), next to the box's pinned hipMemcpyAsync rate and the simple
(unpipelined, runtime-staged) path."""
import os

os.environ.setdefault("BITNUC_PIPE_IMPL", "staged")  # this tool studies the STAGED engine's thread budget / placement (the direct engine ships: tools/ab_pipe_impl.py)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bitnuc_amd

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**9
pin = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
devb = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
for name, (dst, src) in (("pinned H2D", (devb, pin)), ("pinned D2H", (pin, devb))):
    ts = []
    for _ in range(4):
        torch.cuda.synchronize()
        t = time.perf_counter()
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    print(f"{name}: {(1 << 30) / min(ts[1:]) / 1e9:.1f} GB/s")
a = np.empty(1 << 30, dtype=np.uint8)
a[:] = 65
b = np.empty(1 << 30, dtype=np.uint8)
b[:] = 1
t = time.perf_counter()
np.copyto(b, a)
print(f"single-thread host memcpy (numpy, 1 GiB): {(1 << 30) / (time.perf_counter() - t) / 1e9:.1f} GB/s")
del pin, devb, b
seq = np.repeat(np.frombuffer(b"ACGT", dtype=np.uint8)[np.frombuffer(np.random.default_rng(1).bytes(n // 4 + 1), dtype=np.uint8) & 3], 4)[:n].copy()
print(f"cores visible: {len(os.sched_getaffinity(0))}")
from bitnuc_amd import _lib
lib = _lib.load()
for mode, name in ((0, "pageable -> pageable"), (1, "pageable -> pinned"), (2, "pinned -> pageable")):
    print(f"staging pool memcpy, 512 MiB, {name}: " + "  ".join(f"{t} thr {lib.bitnuc_selftime_host_copy(1 << 29, t, mode):5.1f}" for t in (1, 2, 4, 8, 16, 32)) + "  GB/s", flush=True)


w = np.zeros((n + 31) // 32, dtype=np.uint64)  # caller-owned, already touched: a fresh np.empty output is timed by its page faults
back = np.zeros(n, dtype=np.uint8)


def run(label):
    ctx = bitnuc_amd.Context(0)
    te, td = [], []
    for _ in range(4):
        t = time.perf_counter()
        ctx.encode_into(seq, w)
        te.append(time.perf_counter() - t)
        t = time.perf_counter()
        ctx.decode_into(w, n, back)
        td.append(time.perf_counter() - t)
    ok = bool(np.array_equal(back, seq))
    print(f"{label:44s} encode {n / min(te[1:]) / 1e9:5.1f} Gbases/s   decode {n / min(td[1:]) / 1e9:5.1f} Gbases/s   {'ok' if ok else 'MISMATCH'}", flush=True)
    ctx.close()


os.environ["BITNUC_HOST_THREADS"] = "1"
c0 = bitnuc_amd.Context(0)
c0.require_variant("host_pipeline", 0)
te = []
for _ in range(3):
    t = time.perf_counter()
    c0.encode_into(seq, w)
    te.append(time.perf_counter() - t)
print(f"{'simple path (runtime-staged pageable copies)':44s} encode {n / min(te[1:]) / 1e9:5.1f} Gbases/s")
c0.close()
for threads in (4, 12, 16):
    for mb in (32, 128):
        os.environ["BITNUC_HOST_THREADS"] = str(threads)
        os.environ["BITNUC_PIPE_CHUNK_MB"] = str(mb)
        run(f"pipelined, {threads:2d} staging threads, {mb:3d} Mi-base chunks")
