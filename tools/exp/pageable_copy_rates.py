#!/usr/bin/env python3
"""What does the HIP runtime do with PAGEABLE host memory?  hipMemcpyAsync (through torch's copy_) of 1 GB between a numpy buffer
and the device, whole and in pieces, each direction alone.  Context for csrc/host_pipe.h: which side needs the library's own
staging threads."""
import statistics
import time

import numpy as np
import torch

n = 1 << 30
host = np.random.default_rng(1).integers(0, 255, n, dtype=np.uint8)
dst_host = np.zeros(n, dtype=np.uint8)
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
h = torch.from_numpy(host)
hd = torch.from_numpy(dst_host)


def med(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    return statistics.median(ts)


for piece in (n, 256 << 20, 64 << 20, 32 << 20, 8 << 20):
    def h2d():
        for o in range(0, n, piece):
            dev[o:o + piece].copy_(h[o:o + piece], non_blocking=True)

    def d2h():
        for o in range(0, n, piece):
            hd[o:o + piece].copy_(dev[o:o + piece], non_blocking=True)
    a, b = med(h2d), med(d2h)
    print(f"pieces of {piece >> 20:5d} MiB: pageable H2D {n / a / 1e9:6.1f} GB/s   pageable D2H {n / b / 1e9:6.1f} GB/s")
ph = torch.empty(n, dtype=torch.uint8).pin_memory()
a = med(lambda: dev.copy_(ph, non_blocking=True))
b = med(lambda: ph.copy_(dev, non_blocking=True))
print(f"pinned, one piece:      H2D {n / a / 1e9:6.1f} GB/s   D2H {n / b / 1e9:6.1f} GB/s")

# ---- is the pageable copy asynchronous for the host, and do the two directions overlap? ---------------------------------------
import threading

s_in, s_out = torch.cuda.Stream(), torch.cuda.Stream()
dev2 = torch.empty(n, dtype=torch.uint8, device="cuda")
piece = 32 << 20
torch.cuda.synchronize()
t = time.perf_counter()
with torch.cuda.stream(s_in):
    dev[:piece].copy_(h[:piece], non_blocking=True)
t_call = time.perf_counter() - t
torch.cuda.synchronize()
t_all = time.perf_counter() - t
print(f"pageable H2D of 32 MiB: the call returns after {t_call * 1e3:.2f} ms, the copy is done after {t_all * 1e3:.2f} ms")
t = time.perf_counter()
with torch.cuda.stream(s_out):
    hd[:piece].copy_(dev2[:piece], non_blocking=True)
t_call = time.perf_counter() - t
torch.cuda.synchronize()
t_all = time.perf_counter() - t
print(f"pageable D2H of 32 MiB: the call returns after {t_call * 1e3:.2f} ms, the copy is done after {t_all * 1e3:.2f} ms")


def both(threads):
    def a():
        with torch.cuda.stream(s_in):
            for o in range(0, n, piece):
                dev[o:o + piece].copy_(h[o:o + piece], non_blocking=True)
            s_in.synchronize()

    def b():
        with torch.cuda.stream(s_out):
            for o in range(0, n, piece):
                hd[o:o + piece].copy_(dev2[o:o + piece], non_blocking=True)
            s_out.synchronize()
    if threads:
        ta, tb = threading.Thread(target=a), threading.Thread(target=b)
        ta.start(); tb.start(); ta.join(); tb.join()
    else:  # one thread, interleaved
        with torch.cuda.stream(s_in):
            pass
        for o in range(0, n, piece):
            with torch.cuda.stream(s_in):
                dev[o:o + piece].copy_(h[o:o + piece], non_blocking=True)
            with torch.cuda.stream(s_out):
                hd[o:o + piece].copy_(dev2[o:o + piece], non_blocking=True)
        torch.cuda.synchronize()


for threads in (False, True):
    x = med(lambda: both(threads))
    print(f"1 GiB in AND 1 GiB out, pageable both, 32 MiB pieces, {'two host threads' if threads else 'one host thread, interleaved'}: {2 * n / x / 1e9:6.1f} GB/s both directions together")

# ---- the host path's two traffic mixes with direct pageable copies from two host threads (no staging threads, no pinned buffers) --
def mix(n_in, n_out):
    def a():
        with torch.cuda.stream(s_in):
            for o in range(0, n_in, piece):
                dev[o:o + piece].copy_(h[o:o + piece], non_blocking=True)
            s_in.synchronize()

    def b():
        with torch.cuda.stream(s_out):
            po = piece * n_out // n_in if n_out < n_in else piece
            for o in range(0, n_out, po):
                hd[o:o + po].copy_(dev2[o:o + po], non_blocking=True)
            s_out.synchronize()
    ta, tb = threading.Thread(target=a), threading.Thread(target=b)
    ta.start(); tb.start(); ta.join(); tb.join()


for name, n_in, n_out in (("encode mix (1 B in, 0.25 B out per base)", n, n // 4), ("decode mix (0.25 B in, 1 B out per base)", n // 4, n), ("scan mix (1 B in, 1 B out per window)", n, n)):
    if n_in < n_out:
        def run():
            # the smaller side uses proportionally smaller pieces
            def a():
                with torch.cuda.stream(s_in):
                    pi = piece * n_in // n_out
                    for o in range(0, n_in, pi):
                        dev[o:o + pi].copy_(h[o:o + pi], non_blocking=True)
                    s_in.synchronize()

            def b():
                with torch.cuda.stream(s_out):
                    for o in range(0, n_out, piece):
                        hd[o:o + piece].copy_(dev2[o:o + piece], non_blocking=True)
                    s_out.synchronize()
            ta, tb = threading.Thread(target=a), threading.Thread(target=b)
            ta.start(); tb.start(); ta.join(); tb.join()
        x = med(run)
    else:
        x = med(lambda: mix(n_in, n_out))
    print(f"{name}: {max(n_in, n_out) / x / 1e9:6.1f} G items/s  ({(n_in + n_out) / x / 1e9:6.1f} GB/s over the link, both directions)")
