#!/usr/bin/env python3
"""Kernel-variant sweep on one GPU: times every encode/decode variant x grid shape with
HIP events (interleaved rounds in one process), prints a table sorted by GB/s."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitnuc_amd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bases", type=int, default=10**9)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--grids", type=str, default="0,2,4,8,16,32")
    ap.add_argument("--variants", type=str, default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream()
    from bitnuc_amd import build as _b
    ctx = bitnuc_amd.Context(0, stream=stream.cuda_stream, lib_path=_b.ensure_built(sweep=True))  # the evidence build holds all 47 variants
    n = args.bases
    nw = (n + 31) // 32
    R = 3
    seqs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(R)]
    words = [torch.empty(nw, dtype=torch.int64, device=dev) for _ in range(R)]
    backs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(R)]
    for r in range(R):
        ctx.nucgen_dev(seqs[r], n, 0xB17C0DE + r)
        ctx.encode_dev(seqs[r], n, words[r])
    ctx.sync()
    nv = ctx.get("num_variants")
    variants = [int(v) for v in args.variants.split(",")] if args.variants else list(range(nv))
    grids = [int(g) for g in args.grids.split(",")]
    results = {}
    it = 0
    for rnd in range(args.rounds):
        for v in variants:
            for g in grids:
                ctx.require_variant("encode", v)
                ctx.require_variant("decode", v)
                ctx.require_variant("grid_mult", g)
                r = it % R
                it += 1
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record(stream)
                ctx.encode_dev(seqs[r], n, words[r])
                e[1].record(stream)
                ctx.decode_dev(words[r], nw, n, backs[r])
                e[2].record(stream)
                torch.cuda.synchronize()
                results.setdefault(("enc", v, g), []).append(e[0].elapsed_time(e[1]))
                results.setdefault(("dec", v, g), []).append(e[1].elapsed_time(e[2]))
    ctx.sync()
    rows = []
    for (kind, v, g), ms in results.items():
        ms = ms[1:] if len(ms) > 2 else ms
        med, best = statistics.median(ms), min(ms)
        rows.append((kind, 1.25 * n / (med * 1e-3) / 1e9, 1.25 * n / (best * 1e-3) / 1e9, v, g, med))
    for kind in ("enc", "dec"):
        print(f"== {kind}: GB/s(median)  GB/s(best)  variant grid_mult  ms(median)")
        for row in sorted([r for r in rows if r[0] == kind], key=lambda r: -r[1]):
            print(f"{row[1]:9.1f} {row[2]:9.1f}   v{row[3]:<3d} g{row[4]:<3d} {row[5]:.4f}")


if __name__ == "__main__":
    main()
