#!/usr/bin/env python3
"""Bank-conflict model of the decode strip's three ds_or_b32 per word on 150-base reads (32 banks, 32-lane groups, extra
cycles per group = max number of distinct addresses on one bank - 1), for the linear strip and for two split-by-parity
layouts.  All three give 6.0 extra cycles per 64-word tile = what SQ_LDS_BANK_CONFLICT reports for the decode kernels
(profiles/r02_lds_split_strip.txt): a lane's first chunk index advances by 1 or 2 per lane depending on the pads before
it, so the 32 lanes of a group pick an arbitrary half of a 64-chunk window, and no static layout maps every such
pick to 32 different banks (a single colliding pair costs the same extra cycle as a full 2-way conflict).  The encode's
READS are different: every lane reads both halves at an index that advances by 0 or 1 per lane, which the split layout
makes conflict-free by construction (stream_cut in batch_device.h)."""
import numpy as np

L, WPR = 150, 5


def conflicts(instrs, banks=32):
    tot = 0
    for addrs in instrs:
        for g in (addrs[:32], addrs[32:]):
            tot += max(len({int(x) for x in g if x % banks == b}) for b in range(banks)) - 1
    return tot


res = {"linear": 0, "split, odd half at 80 (16 mod 32)": 0, "split, odd half at 96 (0 mod 32)": 0}
N = 2000
for t in range(N):
    w = np.arange(64 * t, 64 * t + 64)
    start = (w // WPR) * L + 32 * (w % WPR)
    lead = start[0] % 16
    bit = 2 * (lead + start - start[0])
    d = bit >> 5
    res["linear"] += conflicts([d, d + 1, d + 2])
    e, odd = bit >> 6, (bit >> 5) & 1
    for off, key in ((80, "split, odd half at 80 (16 mod 32)"), (96, "split, odd half at 96 (0 mod 32)")):
        d0, d1 = e + odd * off, e + odd + (1 - odd) * off
        res[key] += conflicts([d0, d1, d0 + 1])
for k, v in res.items():
    print(f"{k:40s} {v / N:.2f} extra LDS cycles per tile")
