#!/usr/bin/env python3
"""Sustained (back-to-back, no host sync inside a burst) sweep over (encode variant,
decode variant) pairs: what bench.py's timed loop sees.  Prints per-pair step time and
per-kernel GB/s, best first."""
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
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--burst", type=int, default=9)
    ap.add_argument("--enc", type=str, default="1,2,3,5,11,12,19")
    ap.add_argument("--dec", type=str, default="0,1,4,9,13,18,22,24,25,26,28")
    ap.add_argument("--grids", type=str, default="0")
    ap.add_argument("--cold", type=int, default=1, help="1: decode a set encoded R-1 steps earlier (what bench.py does); 0: decode the words just written (Infinity-Cache warm)")
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
    encs = [int(v) for v in args.enc.split(",")]
    decs = [int(v) for v in args.dec.split(",")]
    res = {}
    for rnd in range(args.rounds):
        for g in [int(x) for x in args.grids.split(",")]:
            for ev in encs:
                for dv in decs:
                    ctx.require_variant("encode", ev)
                    ctx.require_variant("decode", dv)
                    ctx.require_variant("grid_mult", g)
                    evs = []
                    for i in range(args.burst):
                        r = i % R
                        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                        e[0].record(stream)
                        ctx.encode_dev(seqs[r], n, words[r])
                        e[1].record(stream)
                        d = (r + 1) % R if args.cold else r  # --cold: decode words written R-1 steps ago (HBM, not Infinity Cache)
                        ctx.decode_dev(words[d], nw, n, backs[d])
                        e[2].record(stream)
                        evs.append(e)
                    torch.cuda.synchronize()
                    enc = statistics.mean(e[0].elapsed_time(e[1]) for e in evs[3:])
                    dec = statistics.mean(e[1].elapsed_time(e[2]) for e in evs[3:])
                    tot = evs[3][0].elapsed_time(evs[-1][2]) / (len(evs) - 3)
                    res.setdefault((ev, dv, g), []).append((tot, enc, dec))
    ctx.sync()
    rows = []
    for (ev, dv, g), v in res.items():
        tot = statistics.median(x[0] for x in v)
        enc = statistics.median(x[1] for x in v)
        dec = statistics.median(x[2] for x in v)
        rows.append((tot, ev, dv, g, enc, dec))
    print("step_ms  enc_v dec_v grid  enc_ms(GB/s)  dec_ms(GB/s)  Gbases/s(2n/step)")
    for tot, ev, dv, g, enc, dec in sorted(rows):
        print(f"{tot:.4f}   e{ev:<3d} d{dv:<3d} g{g:<2d}  {enc:.4f} ({1.25*n/enc/1e6:6.0f})  {dec:.4f} ({1.25*n/dec/1e6:6.0f})  {2*n/tot/1e6:7.0f}")


if __name__ == "__main__":
    main()
