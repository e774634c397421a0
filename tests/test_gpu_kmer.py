"""GPU parity of the k-mer compositions and their alternative formulations (bitnuc_amd/csrc/kmer_device.h, scan_mfma_device.h; configs 3
and 5): every window of a sequence (src/lib.rs:170-173) in both tilings, the scan forms incl. first-invalid-byte order, the coalesced
many-pair hdist (hamming/scalar.rs:11-48), the quad-transpose encode variants, and the rule that bytes before a batch never reach its first
word.  (Filed by component in round 5; tests from test_gpu_round3.py / test_gpu_round4.py unchanged, round 5's added below them.)"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0xB17C0DE


def test_bytes_before_a_batch_never_reach_its_first_word(ctx, oracle):
    """The batch kernels load aligned 16-byte chunks; what precedes the batch's first base inside its first chunk is not the
    batch's (here: bytes that are not bases at all).  Found in round 3: such a byte in the same DWORD as the first bases used
    to spill into their codes through enc4's multiply-add.  Every lead 1..15 x {plan, tables, fixed-length back-to-back},
    device pointers, against the oracle loop."""
    import torch
    import bitnuc_amd as bn
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(99)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    L, count = 75, 300
    body = alpha[rng.integers(0, 4, size=L * count)]
    exp = np.concatenate([oracle.encode(body[i * L:(i + 1) * L]) for i in range(count)])
    wpr = (L + 31) // 32
    for lead in range(0, 16):
        for junk in (ord("N"), 0xFF, ord("\n"), 0x00):
            buf = np.concatenate([np.full(lead, junk, np.uint8), body, np.full(7, junk, np.uint8)])
            hold = torch.zeros(len(buf) + 16, dtype=torch.uint8, device=dev)
            assert hold.data_ptr() % 16 == 0
            hold[:len(buf)] = torch.from_numpy(buf).to(dev)
            off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L + lead
            wo = torch.zeros(count + 1, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            total = ctx.batch_word_offsets_dev(off, count, wo)
            assert total == count * wpr
            w_tab = torch.zeros(total, dtype=torch.int64, device=dev)
            w_plan = torch.zeros(total, dtype=torch.int64, device=dev)
            w_fix = torch.zeros(total, dtype=torch.int64, device=dev)
            plan = bn.BatchPlan(ctx, off, count)
            ctx.encode_batch_dev(hold, off, wo, count, total, w_tab)
            plan.encode_dev(hold, w_plan)
            ctx.encode_fixed_dev(hold.data_ptr() + lead, L, L, count, w_fix)
            ctx.sync()
            for name, w in (("tables", w_tab), ("plan", w_plan), ("fixed", w_fix)):
                assert np.array_equal(w.cpu().numpy().view(np.uint64), exp), (name, lead, junk)
            plan.close()


# ---- every window of a sequence: line-aligned rounds, windows computed where they are stored -------------------------------
@pytest.mark.parametrize("rounds_per_trip", [1, 2, 4])
def test_windows_line_aligned_rounds_vs_oracle(ctx, sweep_ctx, oracle, rounds_per_trip):
    """kmer_slide2_kernel (`for w in seq.windows(k) { as_2bit(w) }`, src/lib.rs:170-173): rounds of 1024 windows whose 30-base
    halo comes from the next round's registers or one extra load; sizes around the 1024 / 1056-byte round and trip
    boundaries, every k class (<= 16, 17..31, 32), first invalid byte incl. the halo positions."""
    import bitnuc_amd as bn
    ctx = ctx if rounds_per_trip == 4 else sweep_ctx  # the product ships 4 rounds per trip; 1 and 2 live in the evidence build
    prev = ctx.set_variant("slide2_rounds", rounds_per_trip)
    assert ctx.get("slide_impl") == 1 and ctx.get("slide2_rounds") == rounds_per_trip
    rng = np.random.default_rng(4242 + rounds_per_trip)
    alpha = np.frombuffer(b"ACGTacgt", dtype=np.uint8)
    try:
        for k in (1, 2, 15, 16, 17, 21, 31, 32):
            for n in (1055, 1056, 1057, 1056 + k - 1, 2047, 2048, 2079, 2080, 2081, 4 * 1024 + 31, 4 * 1024 + 32, 4 * 1024 + 33,
                      5 * 1024 + 40, 8 * 1024 + 32, 9 * 1024 + 500, 200003):
                if n < k:
                    continue
                s = alpha[rng.integers(0, 8, size=n)]
                count = n - k + 1
                assert np.array_equal(ctx.as_2bit_batch(s, k, 1, count), oracle.as_2bit_batch(s, k, 1, count)), (k, n)
        k, n = 31, 50000
        s = alpha[rng.integers(0, 4, size=n)].copy()
        for pos in (0, 15, 16, 1023, 1024, 1025, 1039, 1040, 1055, 1056, 4095, 4096, 4 * 1024 + 31, 20000, n - 1):
            t = s.copy()
            t[pos] = ord("N")
            if pos + 7 < n:
                t[pos + 7] = ord("X")  # a later invalid byte never wins
            with pytest.raises(bn.NucleotideError) as ei:
                ctx.as_2bit_batch(t, k, 1, n - k + 1)
            assert (ei.value.byte, ei.value.index) == (ord("N"), pos), pos
        t = np.concatenate([s, np.frombuffer(b"N", dtype=np.uint8)])  # a byte past the last window is never examined
        assert np.array_equal(ctx.as_2bit_batch(t, k, 1, n - k + 1), oracle.as_2bit_batch(s, k, 1, n - k + 1))
    finally:
        ctx.set_variant("slide2_rounds", prev)


def test_windows_both_formulations_agree_at_scale(sweep_ctx, oracle):
    """10^8 bases, k = 31: the strip kernel of rounds 1-2 (rounds of 992 windows) and the line-aligned kernel write the same
    10^8 - 30 words; spot blocks against the oracle."""
    import torch
    ctx = sweep_ctx
    dev = torch.device("cuda:0")
    n, k = 10**8 + 13, 31
    seq = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.nucgen_dev(seq, n, SEED)
    count = n - k + 1
    a = torch.zeros(count, dtype=torch.int64, device=dev)
    b = torch.zeros(count, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    prev = ctx.set_variant("slide_impl", 0)
    ctx.as_2bit_batch_dev(seq, k, 1, count, a)
    ctx.set_variant("slide_impl", 1)
    ctx.as_2bit_batch_dev(seq, k, 1, count, b)
    ctx.sync()
    ctx.set_variant("slide_impl", prev)
    assert torch.equal(a, b)
    h = seq.cpu().numpy()
    for start in (0, 1024 * 777 - 40, count - 5000):
        exp = oracle.as_2bit_batch(h[start:start + 5000 + k - 1], k, 1, 5000)
        assert np.array_equal(b[start:start + 5000].cpu().numpy().view(np.uint64), exp), start


def test_encode_quad_variants_vs_oracle(sweep_ctx, oracle):
    """encode variants 47..62 (evidence build): 16-byte stores by a register quad transpose (encode_quad_kernel) -- same
    words as the oracle at tailed sizes, first invalid byte with its index."""
    import bitnuc_amd as bn
    ctx = sweep_ctx
    enc0 = ctx.get("encode")
    rng = np.random.default_rng(5150)
    alpha = np.frombuffer(b"ACGTacgt", dtype=np.uint8)
    try:
        for v in range(47, 63):
            assert ctx.set_variant("encode", v) != -2
            for n in (1, 31, 4095, 4096, 4097, 8192 * 2 + 5, 16384 * 4 + 5, 1000003, (1 << 22) + 17):
                s = alpha[rng.integers(0, 8, size=n)]
                assert np.array_equal(ctx.encode_array(s), oracle.encode(s)), (v, n)
            s = alpha[rng.integers(0, 4, size=300000)].copy()
            s[123457] = ord("N")
            s[200000] = ord("X")
            with pytest.raises(bn.NucleotideError) as ei:
                ctx.encode_array(s)
            assert (ei.value.byte, ei.value.index) == (ord("N"), 123457), v
    finally:
        ctx.set_variant("encode", enc0)
    assert ctx.set_variant("encode", 63) == -2


def test_hdist_words_coalesced_kernel_vs_oracle(ctx, sweep_ctx, oracle):
    """Many-pair / one-query hdist_scalar (hamming/scalar.rs:11-48): the coalesced-load kernel (whole 256-word wave tiles, the
    stored bytes gathered from neighbouring lanes) and the four-contiguous-words kernel give the oracle's distances for
    counts around the tile size, every len class, 16- and 8-byte aligned inputs."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(808)
    product = ctx
    for impl in (1, 0):
        ctx = product if impl == 1 else sweep_ctx  # the four-contiguous-words kernel lost its A/B: evidence build only
        prev = ctx.set_variant("hdist_words_impl", impl)
        try:
            for count in (1, 255, 256, 257, 511, 512, 1000, 256 * 37 + 3, 100003):
                for length in (0, 1, 16, 31, 32):
                    a = rng.integers(0, 1 << 63, size=count + 1, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=count + 1, dtype=np.uint64)
                    b = a ^ (rng.integers(0, 1 << 63, size=count + 1, dtype=np.uint64) & rng.integers(0, 1 << 63, size=count + 1, dtype=np.uint64))
                    for shift in (0, 1):  # 16-byte aligned tables, then 8 bytes into them
                        da = torch.from_numpy(a.view(np.int64)).to(dev)[shift:]
                        db = torch.from_numpy(b.view(np.int64)).to(dev)[shift:]
                        n = count + 1 - shift
                        out = torch.full((n + 8,), 0xEE, dtype=torch.uint8, device=dev)
                        torch.cuda.synchronize()
                        ctx.hdist_pairs_dev(da, db, n, length, out)
                        ctx.sync()
                        exp = oracle.hdist_pairs(a[shift:], b[shift:], length)
                        h = out.cpu().numpy()
                        assert np.array_equal(h[:n], exp) and (h[n:] == 0xEE).all(), (impl, count, length, shift)
                        q = int(b[0])
                        ctx.hdist_query_dev(q, da, n, length, out)
                        ctx.sync()
                        assert np.array_equal(out.cpu().numpy()[:n], oracle.hdist_pairs(a[shift:], np.full(n, q, dtype=np.uint64), length)), (impl, count, length, shift)
        finally:
            ctx.set_variant("hdist_words_impl", prev)


@pytest.mark.parametrize("impl", [2, 3, 5], ids=["chunks12", "chunks20", "chunks32"])
def test_scan3_first_invalid_byte_and_later_bytes(sweep_ctx, oracle, impl):
    """kmer_scan3_kernel (a wave owns 12 / 20 / 32 consecutive rounds and carries the halo planes): the first invalid byte wins at round,
    trip and chunk boundaries and inside the halo positions; a byte after the last window is never examined; the shipped form gives
    the same answers (hamming/scalar.rs:11-48 over naive.rs:3-20 per window)."""
    import bitnuc_amd as bn
    ctx = sweep_ctx
    rng = np.random.default_rng(77 + impl)
    alpha = np.frombuffer(b"ACGTacgt", dtype=np.uint8)
    C = {2: 12, 3: 20, 4: 16, 5: 32}[impl]
    n = (2 * C + 5) * 1024 + 77
    s = alpha[rng.integers(0, 8, size=n)].copy()
    k, q = 31, 0x0123456789ABCDEF & ((1 << 62) - 1)
    prev = ctx.set_variant("scan_impl", impl)
    try:
        assert ctx.get("scan_impl") == impl
        assert np.array_equal(ctx.kmer_hdist_scan(s, k, q), oracle.kmer_hdist_scan(s, k, q))
        for pos in (0, 15, 16, 1023, 1024, 1025, 1039, 1040, 1055, 1056, 4095, 4096, 4 * 1024 + 31, C * 1024 - 1, C * 1024, C * 1024 + 17, C * 1024 + 31, C * 1024 + 32,
                    2 * C * 1024 - 1, 2 * C * 1024 + 1, (2 * C + 4) * 1024 + 5, n - k - 1, n - 1):
            t = s.copy()
            t[pos] = ord("N")
            if pos + 9 < n:
                t[pos + 9] = ord("X")  # a later invalid byte never wins
            with pytest.raises(bn.NucleotideError) as ei:
                ctx.kmer_hdist_scan(t, k, q)
            assert (ei.value.byte, ei.value.index) == (ord("N"), pos), pos
        for kk in (1, 2, 16, 17, 32):
            for m in (kk, 1056, 1057, C * 1024 + 31, C * 1024 + 32, C * 1024 + 33, n):
                if m < kk:
                    continue
                assert np.array_equal(ctx.kmer_hdist_scan(s[:m], kk, q), oracle.kmer_hdist_scan(s[:m], kk, q)), (kk, m)
    finally:
        ctx.set_variant("scan_impl", prev)


# ---- SURVEY 8e's other shards on the HIP path: scan (with its halo), decode and the dense k-mer batch, shard by shard ------------------------
@pytest.mark.parametrize("world", [2, 3, 8])
def test_scan_decode_and_kmer_batch_shards_concatenate_on_the_hip_path(oracle, world):
    """"Decode, k-mer batch and scan shard the same way (scan needs a 30-base halo per shard)" -- run on the kernels, not only in dist.py's
    arithmetic: every rank's slice goes through its OWN Context (own stream) on device 0, writing its part of one output in place;
    concatenation == the unsharded launch == the oracle.  Sizes: one where a shard is shorter than a 1 056-byte round (the whole shard is the
    kernels' tail path) and one where shard boundaries fall in the middle of rounds and the last shard ends ragged.  No collective anywhere
    (window idiom: src/lib.rs:170-173; per-word independence of decode: unpacking/avx.rs:137-141; no carry between words: packing/avx.rs:138-145)."""
    import torch
    import bitnuc_amd
    from bitnuc_amd.dist import scan_shard_range, shard_range
    dev = torch.device("cuda:0")
    ranks = [bitnuc_amd.Context(0) for _ in range(world)]
    whole = bitnuc_amd.Context(0)
    for c in ranks + [whole]:
        c.set_variant("force_gpu", 1)
    try:
        k, q = 31, 0x2B1B4E1B1B1B1B1B & ((1 << 62) - 1)
        for n in (world * 700 + 30, 100003, 3 * 10**6 + 77):
            seq = oracle.nucgen(n, SEED + n, flags=2)  # lower-case mix
            t = torch.from_numpy(seq).to(dev)
            # -- scan: rank r reads bases [first, first + count + k - 1) and writes windows [first, first + count)
            want = oracle.kmer_hdist_scan(seq, k, q)
            d_sh = torch.full((n,), 0xEE, dtype=torch.uint8, device=dev)
            d_one = torch.full((n,), 0xEE, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            covered = 0
            for r, c in enumerate(ranks):
                first, count, nread = scan_shard_range(n, k, r, world)
                assert first == covered and first % 32 == 0
                covered += count
                if count:
                    c.kmer_hdist_scan_dev(t.data_ptr() + first, nread, k, q, d_sh.data_ptr() + first)
            whole.kmer_hdist_scan_dev(t, n, k, q, d_one)
            for c in ranks + [whole]:
                c.sync()
            assert covered == n - k + 1
            got = d_sh.cpu().numpy()
            assert np.array_equal(got[:covered], want) and bool((got[covered:] == 0xEE).all()), (world, n)
            assert torch.equal(d_sh, d_one), (world, n)
            # -- decode: words [a / 32, ceil(b / 32)) -> bases [a, b)
            words = oracle.encode(seq)
            w = torch.from_numpy(words.view(np.int64).copy()).to(dev)
            b_sh = torch.full((n + 16,), 0xEE, dtype=torch.uint8, device=dev)
            b_one = torch.full((n + 16,), 0xEE, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            for r, c in enumerate(ranks):
                a, b = shard_range(n, r, world)
                if b > a:
                    c.decode_dev(w.data_ptr() + 8 * (a // 32), (b - a + 31) // 32, b - a, b_sh.data_ptr() + a)
            whole.decode_dev(w, words.size, n, b_one)
            for c in ranks + [whole]:
                c.sync()
            assert torch.equal(b_sh, b_one) and np.array_equal(b_sh.cpu().numpy()[:n], oracle.decode(words, n)) and bool((b_sh[n:] == 0xEE).all()), (world, n)
            # -- dense k-mer batch (stride == k): the k-mers split into contiguous runs
            count = n // k
            want_k = oracle.as_2bit_batch(seq, k, k, count)
            o_sh = torch.full((count + 2,), -1, dtype=torch.int64, device=dev)
            o_one = torch.full((count + 2,), -1, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            for r, c in enumerate(ranks):
                a, b = shard_range(count, r, world)  # k-mer indices
                if b > a:
                    c.as_2bit_batch_dev(t.data_ptr() + a * k, k, k, b - a, o_sh.data_ptr() + 8 * a)
            whole.as_2bit_batch_dev(t, k, k, count, o_one)
            for c in ranks + [whole]:
                c.sync()
            assert torch.equal(o_sh, o_one) and np.array_equal(o_sh.cpu().numpy()[:count].view(np.uint64), want_k) and int(o_sh[count]) == -1, (world, n)
        # an invalid byte in the scan's halo region belongs to BOTH neighbouring shards' reads: the lower rank reports it at its own offset
        n = 100003
        seq = oracle.nucgen(n, SEED, flags=0).copy()
        first1, count1, _ = scan_shard_range(n, k, 1, world)
        seq[first1 + 5] = ord("N")  # inside rank 0's halo and rank 1's first window
        t = torch.from_numpy(seq).to(dev)
        d = torch.zeros(n, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        for r in (0, 1):
            first, count, nread = scan_shard_range(n, k, r, world)
            ranks[r].kmer_hdist_scan_dev(t.data_ptr() + first, nread, k, q, d.data_ptr() + first)
            with pytest.raises(bitnuc_amd.NucleotideError) as ei:
                ranks[r].sync()
            assert (ei.value.byte, ei.value.index) == (ord("N"), first1 + 5 - first), r  # shard-relative, like the sharded encode
    finally:
        for c in ranks + [whole]:
            c.close()
