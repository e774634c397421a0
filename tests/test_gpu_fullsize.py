"""GPU parity at BASELINE size: 10^9 + 17 bases with the lower-case mix, the benches' cyclic pattern
(benches/simd_comparison.rs:4-7) and planted invalid bytes (packing/mod.rs:181-196, packing/avx.rs:86-91,142-143); every output element of
configs 3 and 5 (10^8 dense 31-mers against an independent closed form, 10^9 - 30 window distances against the oracle run on the host
cores).  (Filed by component in round 5; tests from test_gpu_round2.py / test_gpu_round4.py unchanged.)"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0xB17C0DE


def gen_words(torch, n, seed, dev):
    """Closed form of the seeded stream's packed words (the generator's words ARE the packed words)."""
    nw = (n + 31) // 32
    idx = torch.arange(1, nw + 1, dtype=torch.int64, device=dev)
    z = idx * (-7046029254386353131) + seed  # 0x9E3779B97F4A7C15 as i64, wraps mod 2^64

    def lsr(x, s):
        return (x >> s) & ((1 << (64 - s)) - 1)
    z = (z ^ lsr(z, 30)) * (-4658895280553007687)   # 0xBF58476D1CE4E5B9
    z = (z ^ lsr(z, 27)) * (-7723592293110705685)   # 0x94D049BB133111EB
    z = z ^ lsr(z, 31)
    if n % 32:
        z[-1] &= (1 << (2 * (n % 32))) - 1
    return z


# ---- BASELINE configs 3 and 5, every output element ---------------------------------------------------------------------------
def _lsr(x, s):  # logical shift right on int64 tensors (s: int or tensor, 1 <= s <= 63)
    return (x >> s) & ~(torch_min_i64() >> (s - 1))


def torch_min_i64():
    import torch
    return torch.tensor(-(1 << 63), dtype=torch.int64, device="cuda:0")


def _generator_words(first, count, seed):
    """Words first .. first+count of the seeded stream's 2-bit encoding: base i of the stream IS field i % 32 of
    splitmix64(seed + (i / 32 + 1) * 0x9E3779B97F4A7C15) (include/bitnuc_hip.h, bitnuc_nucgen_dev), so the stream's packed form is
    that word sequence -- computed here with torch integer arithmetic, independently of every kernel of the library."""
    import torch
    idx = torch.arange(first + 1, first + count + 1, dtype=torch.int64, device="cuda:0")
    z = idx * (-7046029254386353131) + seed        # 0x9E3779B97F4A7C15 as i64, wraps mod 2^64
    z = (z ^ _lsr(z, 30)) * (-4658895280553007687)  # 0xBF58476D1CE4E5B9
    z = (z ^ _lsr(z, 27)) * (-7723592293110705685)  # 0x94D049BB133111EB
    return z ^ _lsr(z, 31)


def test_nucgen_lowercase_and_cyclic_flags(ctx, oracle):
    import torch
    dev = torch.device("cuda:0")
    for n, first, flags in [(1000, 0, 2), (100003, 32 * 77, 2), (4097, 5, 2), (333, 1 << 40, 2), (1000, 3, 3), (70, 17, 3), (65, 31, 2)]:
        t = torch.zeros(n + 16, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ctx.nucgen_dev(t, n, SEED, first, flags)
        ctx.sync()
        h = t.cpu().numpy()
        exp = oracle.nucgen(n, SEED, first, flags)
        assert np.array_equal(h[:n], exp), (n, first, flags)
        assert not h[n:].any()
    big = oracle.nucgen(1 << 20, SEED, 0, 2)
    frac = float((big & 0x20).astype(bool).mean())
    assert 0.24 < frac < 0.26  # p = 0.25
    assert set(np.unique(big)) == set(b"ACGTacgt")


def test_full_size_lowercase_mix_and_cyclic(ctx, oracle):
    import torch
    dev = torch.device("cuda:0")
    n = 10**9 + 17
    nw = (n + 31) // 32
    seq = torch.empty(n, dtype=torch.uint8, device=dev)
    words = torch.empty(nw, dtype=torch.int64, device=dev)
    back = torch.empty(n, dtype=torch.uint8, device=dev)
    # (1) lower-case mix p = 0.25: same words as the upper-case stream (as_2bit("acgt") == as_2bit("ACGT"), packing/mod.rs:181),
    #     decode gives the upper-cased input
    ctx.nucgen_dev(seq, n, SEED, 0, 2)
    torch.cuda.synchronize()
    ctx.encode_dev(seq, n, words)
    ctx.decode_dev(words, nw, n, back)
    ctx.sync()
    assert torch.equal(words, gen_words(torch, n, SEED, dev))
    assert torch.equal(back, seq & 0xDF)
    lower = int(((seq & 0x20) != 0).sum().item())
    assert 0.2499 < lower / n < 0.2501
    for off in (0, 32 * 12_345_678, (n // 32 - 4096) * 32):  # spot blocks against the CPU oracle, bytes as generated
        m = min(32 * 4096, n - off)
        h = seq[off:off + m].cpu().numpy()
        assert np.array_equal(h, oracle.nucgen(m, SEED, off, 2))
        assert np.array_equal(words[off // 32: off // 32 + (m + 31) // 32].cpu().numpy().view(np.uint64), oracle.encode(h))
    # (2) the benches' cyclic pattern: every full word is 0xE4E4..E4, the 17-base tail word its low 34 bits
    ctx.nucgen_dev(seq, n, SEED, 0, 1)
    torch.cuda.synchronize()
    ctx.encode_dev(seq, n, words)
    ctx.decode_dev(words, nw, n, back)
    ctx.sync()
    e4 = int(np.uint64(0xE4E4E4E4E4E4E4E4).view(np.int64))
    assert bool((words[:-1] == e4).all().item())
    assert int(words[-1].item()) == 0xE4E4E4E4E4E4E4E4 & ((1 << 34) - 1)
    assert torch.equal(back, seq)
    assert bytes(seq[:8].cpu().numpy()) == b"ACGTACGT" and bytes(seq[-5:].cpu().numpy()) == bytes(b"ACGT"[(n - 5 + i) % 4] for i in range(5))


def test_full_size_planted_invalid_bytes(ctx, oracle):
    import torch
    import bitnuc_amd as bn
    dev = torch.device("cuda:0")
    n = 10**9 + 17
    nw = (n + 31) // 32
    seq = torch.empty(n, dtype=torch.uint8, device=dev)
    words = torch.empty(nw, dtype=torch.int64, device=dev)
    ctx.nucgen_dev(seq, n, SEED)
    ctx.sync()
    expect = gen_words(torch, n, SEED, dev)
    for pos in (0, 15, 16, 31, 32, n - 1):
        orig = int(seq[pos].item())
        for byte in (ord("N"), 0x00, 0xFF):
            seq[pos] = byte
            later = pos + 1000 if pos + 1000 < n else None
            if later is not None:
                keep = int(seq[later].item())
                seq[later] = ord("X")  # a later invalid byte must not be the one reported
            words.fill_(-1)
            torch.cuda.synchronize()
            ctx.encode_dev(seq, n, words)
            with pytest.raises(bn.NucleotideError) as ei:
                ctx.sync()
            assert (ei.value.kind, ei.value.byte, ei.value.index) == ("InvalidBase", byte, pos), (pos, byte)
            # the reference's Vec holds the words of the chunks before the failing one (packing/avx.rs:142-143)
            k = pos // 32
            assert torch.equal(words[:k], expect[:k]), (pos, byte)
            if later is not None:
                seq[later] = keep
        seq[pos] = orig
    torch.cuda.synchronize()
    ctx.encode_dev(seq, n, words)
    ctx.sync()
    assert torch.equal(words, expect)


def test_config3_every_dense_31mer_against_the_closed_form(ctx, oracle):
    """BASELINE config 3 at full size: 10^8 back-to-back 31-mers (naive.rs:3-20 per k-mer).  The input is the seeded stream, whose
    2-bit encoding is the generator's own word sequence; k-mer j is bits [62 j, 62 j + 62) of that bit stream.  ALL 10^8 output
    words are compared with that closed form (three blocks of 5 000 were compared before), in chunks of 2^24."""
    import torch
    dev = torch.device("cuda:0")
    count, k = 10**8, 31
    seq = torch.empty(count * k, dtype=torch.uint8, device=dev)
    ctx.nucgen_dev(seq, count * k, SEED)
    out = torch.empty(count, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.as_2bit_batch_dev(seq, k, k, count, out)
    ctx.sync()
    # pin the closed form itself to the oracle on one block (generator words == oracle encode of the oracle's stream)
    assert np.array_equal(_generator_words(5_000_000, 4096, SEED).cpu().numpy().view(np.uint64), oracle.encode(oracle.nucgen(32 * 4096, SEED, 32 * 5_000_000)))
    CH = 1 << 24
    compared = 0
    for j0 in range(0, count, CH):
        m = min(CH, count - j0)
        w_first = (62 * j0) >> 6
        w_count = ((62 * (j0 + m) + 63) >> 6) - w_first + 1  # one word beyond: the last k-mer's high part may be empty
        W = _generator_words(w_first, w_count, SEED)
        bit = torch.arange(j0, j0 + m, dtype=torch.int64, device=dev) * 62
        wi = (bit >> 6) - w_first
        sh = bit & 63
        lo = torch.where(sh == 0, W[wi], _lsr(W[wi], torch.clamp(sh, min=1)))
        hi = torch.where(sh <= 2, torch.zeros_like(lo), W[wi + 1] << ((64 - sh) & 63))
        expect = (lo | hi) & ((1 << 62) - 1)
        if not torch.equal(out[j0:j0 + m], expect):
            bad = int((out[j0:j0 + m] != expect).nonzero()[0]) + j0
            h = seq[bad * k:(bad + 1) * k].cpu().numpy()
            raise AssertionError(f"k-mer {bad}: kernel {int(out[bad]) & (2**64 - 1):#x}, closed form {int(expect[bad - j0]) & (2**64 - 1):#x}, oracle {oracle.as_2bit(h):#x}")
        compared += m
    assert compared == count
    # and the closed form agrees with the oracle's per-k-mer loop where the old test looked
    for j0 in (0, 12_345_678, count - 5000):
        h = seq[j0 * k:(j0 + 5000) * k].cpu().numpy()
        assert np.array_equal(out[j0:j0 + 5000].cpu().numpy().view(np.uint64), oracle.as_2bit_batch(h, k, k, 5000))


def test_config5_every_window_of_the_scan_against_the_oracle(ctx, oracle):
    """BASELINE config 5 at full size: 10^9 bases, k = 31, one query: ALL 10^9 - 30 distances -- and the fused count of d <= tau for nine thresholds -- against the oracle's loop
    (naive.rs:3-20 per window, then hamming/scalar.rs:11-48) run over the whole input on the host cores -- 32-window-aligned slices
    from bitnuc_amd.dist.scan_shard_range, one thread each (the oracle is the checker here, nothing of it is timed or shipped).
    Compared by a 64-bit sum per 1 MiB block, with a full compare of a block that differs."""
    import ctypes as C
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from bitnuc_amd.dist import scan_shard_range
    dev = torch.device("cuda:0")
    n, k = 10**9, 31
    nwin = n - k + 1
    ref = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.nucgen_dev(ref, n, SEED)
    ctx.sync()
    qpos = 777_777_777
    q = oracle.as_2bit(ref[qpos:qpos + k].cpu().numpy())
    dist = torch.empty(nwin, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.kmer_hdist_scan_dev(ref, n, k, q, dist)
    ctx.sync()
    assert int(dist[qpos]) == 0 and int(dist.max()) <= k
    # ... and the fused count (three channels per base, the threshold inside the product) for thresholds from "almost nothing" to "everything"
    taus = (0, 8, 16, 20, 23, 26, 30, 31, 40)
    cnt = torch.zeros(len(taus), dtype=torch.int64, device=dev)
    for i, tau in enumerate(taus):
        ctx.kmer_hdist_count_dev(ref, n, k, q, tau, cnt[i:])
    ctx.sync()
    got_counts = [int(x) for x in cnt.cpu()]
    h_ref = ref.cpu().numpy()
    assert np.array_equal(h_ref[:1 << 20], oracle.nucgen(1 << 20, SEED))  # the input is the oracle's stream too
    h_got = dist.cpu().numpy()
    del ref, dist
    h_exp = np.zeros(nwin, dtype=np.uint8)
    h_exp[::4096] = 1  # touch the pages before the threads do
    lib = oracle.lib()
    threads = max(1, min(32, len(os.sched_getaffinity(0))))
    parts = 8 * threads

    def run(r):
        first, cnt, nread = scan_shard_range(n, k, r, parts)
        if cnt == 0:
            return 0
        e = oracle.OrcErr()
        st = lib.orc_kmer_hdist_scan(C.c_void_p(h_ref.ctypes.data + first), nread, k, C.c_uint64(q), C.c_void_p(h_exp.ctypes.data + first), C.byref(e))
        assert st == 0, (r, st)
        return cnt
    with ThreadPoolExecutor(threads) as ex:
        done = sum(ex.map(run, range(parts)))
    assert done == nwin
    # 64-bit sum per 1 MiB block (as u64 lanes of 8 distances: carries cannot cancel a difference within a lane pair by accident the
    # way a byte sum could), full compare where a block differs
    BLK = 1 << 20
    whole = nwin // BLK * BLK
    sums_got = h_got[:whole].view(np.uint64).reshape(-1, BLK // 8).sum(axis=1, dtype=np.uint64)
    sums_exp = h_exp[:whole].view(np.uint64).reshape(-1, BLK // 8).sum(axis=1, dtype=np.uint64)
    badblocks = np.nonzero(sums_got != sums_exp)[0]
    for b in badblocks[:1]:
        i = int(np.nonzero(h_got[b * BLK:(b + 1) * BLK] != h_exp[b * BLK:(b + 1) * BLK])[0][0]) + int(b) * BLK
        raise AssertionError(f"window {i}: kernel {h_got[i]}, oracle {h_exp[i]} ({len(badblocks)} of {whole // BLK} blocks differ)")
    assert np.array_equal(h_got[whole:], h_exp[whole:])
    assert np.array_equal(h_got, h_exp)  # a 1 GB memcmp is cheap: the block sums above only localise a failure
    below = np.cumsum(np.bincount(h_exp, minlength=k + 1))  # below[t] = windows with d <= t, from the ORACLE's distances
    assert got_counts == [int(below[min(t, k)]) for t in taus], (got_counts, [int(below[min(t, k)]) for t in taus])
    assert got_counts[0] >= 1 and got_counts[-1] == nwin
