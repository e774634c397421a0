"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the
C ABI, against the CPU oracle on identical inputs, against the reference's golden
vectors, and -- at BASELINE sizes -- through size-independent properties.
Bar: bit-exact (integer / byte work)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RNG = np.random.default_rng(0xB17C0DE)
ALPHA = np.frombuffer(b"ACGT", dtype=np.uint8)
ALPHA8 = np.frombuffer(b"ACGTacgt", dtype=np.uint8)


def rand_seq(n, alpha=ALPHA8):
    return alpha[RNG.integers(0, len(alpha), size=n)]


# ---- golden vectors through the reference-named API -----------------------------------
def test_golden_as_2bit(ctx, golden):
    import bitnuc_amd as bn
    for v in golden["as_2bit"]:
        assert ctx.as_2bit(v["seq"].encode()) == v["packed"], v["src"]
    ci = golden["as_2bit_case_insensitive"]
    assert ctx.as_2bit(ci["lower"].encode()) == ctx.as_2bit(ci["upper"].encode())
    assert ctx.as_2bit(b"") == 0
    for v in golden["as_2bit_err"]:
        with pytest.raises(bn.NucleotideError) as ei:
            ctx.as_2bit(v["seq"].encode())
        assert ei.value.kind == v["status"], v["src"]
        if "byte" in v:
            assert ei.value.byte == v["byte"]
        if "value" in v:
            assert ei.value.len == v["value"]
    assert bn.NucleotideError("InvalidBase", byte=ord("N")) == pytest.raises(bn.NucleotideError, ctx.as_2bit, b"ACGN").value


def test_golden_from_2bit(ctx, golden):
    import bitnuc_amd as bn
    unpacked = bytearray()
    for v in golden["from_2bit"]:
        ctx.from_2bit(v["packed"], v["n"], unpacked)
        assert bytes(unpacked) == v["seq"].encode(), v["src"]
        unpacked.clear()
    for v in golden["from_2bit_err"]:
        with pytest.raises(bn.NucleotideError) as ei:
            ctx.from_2bit(v["packed"], v["n"], bytearray())
        assert ei.value.kind == v["status"] and ei.value.len == v["value"]
    for v in golden["from_2bit_append"]:  # append semantics, unpacking/avx.rs:185-194
        p = ctx.as_2bit(v["seq"].encode())
        obs = bytearray()
        for _ in range(v["calls"]):
            ctx.from_2bit(p, v["n"], obs)
        assert bytes(obs) == v["expected"].encode()
    rp = golden["roundtrip_prefixes"]
    for n in range(rp["lens"][0], rp["lens"][1] + 1):
        b = rp["seq"].encode()[:n]
        assert ctx.from_2bit_alloc(ctx.as_2bit(b), n) == b
    for s in golden["roundtrip_strings"]["cases"]:
        assert ctx.from_2bit_alloc(ctx.as_2bit(s.encode()), len(s)) == s.encode()


def test_golden_roundtrip_all_lengths(ctx, golden, oracle):
    # src/utils/mod.rs:113-133 -- encode/decode round trip for every length 1..=1000
    lo, hi = golden["roundtrip_lengths"]["lens"]
    for n in range(lo, hi + 1):
        s = rand_seq(n, ALPHA)
        ebuf = [123]  # encode clears the Vec first
        ctx.encode(s, ebuf)
        assert len(ebuf) == (n + 31) // 32
        assert np.array_equal(np.array(ebuf, dtype=np.uint64), oracle.encode(s))
        dbuf = bytearray(b"xy")  # decode appends
        ctx.decode(ebuf, n, dbuf)
        assert bytes(dbuf) == b"xy" + s.tobytes()


def test_golden_hdist(ctx, golden, oracle):
    import bitnuc_amd as bn
    for v in golden["hdist_scalar"]:
        assert ctx.hdist_scalar(v["u"], v["v"], v["len"]) == v["d"], v["src"]
    for a, b, d in golden["hdist_scalar_strings"]["cases"]:
        assert ctx.hdist_scalar(ctx.as_2bit(a.encode()), ctx.as_2bit(b.encode()), len(a)) == d
    for v in golden["hdist_scalar_err"]:
        with pytest.raises(bn.NucleotideError) as ei:
            ctx.hdist_scalar(v["u"], v["v"], v["len"])
        assert ei.value.kind == v["status"] and ei.value.len == v["value"]
    for v in golden["hdist_err"]:
        with pytest.raises(bn.NucleotideError) as ei:
            ctx.hdist([0] * v["na"], [0] * v["nb"], v["n_bases"])
        assert ei.value.kind == v["status"] and ei.value.len == v["value"]
    for v in golden["hdist"]:
        a, b = ctx.encode_alloc(v["seq1"].encode()), ctx.encode_alloc(v["seq2"].encode())
        assert ctx.hdist(a, b, len(v["seq1"])) == v["d"], v["src"]
    lo, hi = golden["hdist_A_vs_T"]["lens"]
    for n in range(lo, hi + 1):
        assert ctx.hdist(ctx.encode_alloc(b"A" * n), ctx.encode_alloc(b"T" * n), n) == n
    for v in golden["hdist_cyclic"]:
        s1 = np.array([ALPHA[i % v["mod1"]] for i in range(v["l"])], dtype=np.uint8)
        s2 = np.array([ALPHA[i % v["mod2"]] for i in range(v["l"])], dtype=np.uint8)
        assert ctx.hdist(ctx.encode_alloc(s1), ctx.encode_alloc(s2), v["l"]) == int((s1 != s2).sum())


def test_kmer_count_doc_example(ctx, golden):
    v = golden["kmer_count"]
    seq = v["seq"].encode()
    words = ctx.as_2bit_batch(seq, v["k"], stride=1)
    assert int((words == ctx.as_2bit(v["kmer"].encode())).sum()) == v["count"]


# ---- bulk encode / decode vs oracle, every kernel variant ------------------------------
SIZES = [1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 1000, 4095, 4096, 4097, 16384 * 4 + 5, 1000003, (1 << 22) + 17]


@pytest.mark.parametrize("variant", range(47))
def test_encode_decode_variants_vs_oracle(sweep_ctx, oracle, variant):
    ctx = sweep_ctx  # the evidence build holds all 47; the product ships 4 of them (next test)
    assert ctx.get("num_variants") == 47 and ctx.get("sweep_build") == 1
    enc0 = ctx.set_variant("encode", variant)
    dec0 = ctx.set_variant("decode", variant)
    try:
        for n in SIZES:
            s = rand_seq(n)
            w = ctx.encode_array(s)
            exp = oracle.encode(s)
            assert np.array_equal(w, exp), (variant, n)
            d = ctx.decode_array(w, n)
            assert np.array_equal(d, oracle.decode(exp, n)), (variant, n)
            assert bytes(d) == bytes(s).upper()
    finally:
        ctx.set_variant("encode", enc0)
        ctx.set_variant("decode", dec0)


def test_decode_x2_variants_vs_oracle(sweep_ctx, oracle):
    """decode variants 47..54: 8-byte loads + LDS transpose (decode_x2_kernel) for the whole 2 KiB tiles, default kernel for the tail."""
    ctx = sweep_ctx
    dec0 = ctx.get("decode")
    try:
        for v in range(47, 55):
            assert ctx.set_variant("decode", v) != -2
            for n in SIZES + [2047, 2048, 2049, 4096 * 5 + 31, (1 << 21) + 2048 + 17]:
                s = rand_seq(n)
                w = oracle.encode(s)
                assert np.array_equal(ctx.decode_array(w, n), oracle.decode(w, n)), (v, n)
    finally:
        ctx.set_variant("decode", dec0)
    assert ctx.set_variant("decode", 55) == -2


def test_product_ships_only_the_variants_in_use(ctx, oracle):
    assert ctx.get("sweep_build") == 0
    built = [v for v in range(47) if ctx.set_variant("encode", v) != -2]
    ctx.set_variant("encode", 39)
    assert built == [0, 3, 22, 39]
    assert ctx.set_variant("encode", 100) == -2 and ctx.get("encode") == 39  # the ballot formulation is evidence, not product
    assert all(ctx.set_variant("decode", v) == -2 for v in range(47, 56)) and ctx.get("decode") == 22  # decode_x2_kernel: evidence build
    # the other formulations that lost their A/B are evidence too: the product holds one form of each kernel
    for key, shipped, others in (("plan_tiles", 1, (2, 4)), ("plan_enc_tiles", 1, (2, 4)), ("plan_store", 2, (0, 1)), ("fixed_dec_strip", 2, (0, 1)),
                                 ("slide_rounds", 1, (2, 4, 8)), ("slide2_rounds", 4, (1, 2)), ("slide_impl", 1, (0,)), ("batch_tables_impl", 1, (0,)), ("scan_impl", 8, (0, 1, 2, 6, 7)), ("scan_unroll", 4, (1, 2)), ("scan_policy", 3, (0, 1, 2)),
                                 ("scan_mfma_shift", 4, (0, 1, 2, 3, 5)), ("scan_mfma_pack", 1, (0, 2)), ("scan_mfma_unroll", 4, (2, 3)), ("scan_mfma_persist", 0, (1,)), ("scan_mfma_count_persist", 1, (0,)), ("scan_mfma_block", 64, (128, 256)), ("scan_mfma_ch3", 0, (1,)), ("scan_mfma_match", 0, (1,)), ("scan_mfma_count_form", 2, (0, 1)), ("scan_mfma_count_emit", 2, (0, 1)), ("scan_mfma_count_rounds", 4, (2, 3)), ("scan_mfma_count_grid", 12, (4, 18)),
                                 ("dense_unroll", 1, (2, 4)), ("dense_policy", 3, (0, 1, 2)), ("batch_abl", 0, (1,)),
                                 ("batch_dense", 1, (0,)), ("batch_slide", 1, (0,)), ("batch_host_plan", 1, (0,)), ("fixed_stream", 1, (0,)), ("owner_est", 3, (0, 1, 2)),
                                 ("plan_enc_block", 256, (64, 128)), ("kmer_block", 256, (64, 128)), ("hdist_tiled", 0, (1,)), ("hdist_words_impl", 1, (0,)),
                                 ("plan_enc_abl", 0, (1,)), ("dyn_lds", 0, (1024,))):
        assert ctx.get(key) == shipped, key
        assert all(ctx.set_variant(key, v) == -2 for v in others) and ctx.get(key) == shipped, key
    for v in built:
        enc0, dec0 = ctx.set_variant("encode", v), ctx.set_variant("decode", v)
        try:
            for n in SIZES:
                s = rand_seq(n)
                w = ctx.encode_array(s)
                assert np.array_equal(w, oracle.encode(s)), (v, n)
                assert np.array_equal(ctx.decode_array(w, n), oracle.decode(w, n)), (v, n)
        finally:
            ctx.set_variant("encode", enc0)
            ctx.set_variant("decode", dec0)


@pytest.mark.parametrize("grid_mult", [0, 1, 8])
def test_grid_shapes(ctx, oracle, grid_mult):
    prev = ctx.set_variant("grid_mult", grid_mult)
    try:
        for n in [5, 70000, 3000017]:
            s = rand_seq(n)
            w = ctx.encode_array(s)
            assert np.array_equal(w, oracle.encode(s))
            assert np.array_equal(ctx.decode_array(w, n), oracle.decode(w, n))
    finally:
        ctx.set_variant("grid_mult", prev)


def test_decode_ignores_bits_above_n(ctx, oracle):
    w = RNG.integers(0, 1 << 63, size=40, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    for n in [1, 33, 40 * 32 - 1, 40 * 32 - 31, 40 * 32]:
        assert np.array_equal(ctx.decode_array(w, n), oracle.decode(w, n))


def test_invalid_base_first_in_sequence_order(ctx, oracle):
    import bitnuc_amd as bn
    for n in [20, 33, 1000, 70001]:
        for pos in sorted(p for p in {0, 15, 16, 31, 32, n // 2, n - 17 if n > 17 else 0, n - 1} if p < n):
            for bad in (ord("N"), 0x00, 0xFF, ord("@"), ord("B"), ord("u")):
                s = rand_seq(n).copy()
                s[pos] = bad
                if pos + 40 < n:
                    s[pos + 40] = ord("X")  # a later invalid byte must not win
                with pytest.raises(oracle.OracleError) as oe:
                    oracle.encode(s)
                ebuf = []
                with pytest.raises(bn.NucleotideError) as ge:
                    ctx.encode(s, ebuf)
                assert (ge.value.kind, ge.value.byte, ge.value.index) == ("InvalidBase", oe.value.byte, oe.value.index) == ("InvalidBase", bad, pos)
                # the Vec holds the words of the chunks before the failing one (avx.rs:142)
                assert np.array_equal(np.array(ebuf, dtype=np.uint64), oe.value.words)
    # a context keeps working after an error
    assert ctx.as_2bit(b"ACGT") == 0xE4


def test_all_256_byte_values(ctx):
    import bitnuc_amd as bn
    valid = {b: i for i, ch in enumerate(b"ACGT") for b in (ch, ch | 0x20)}
    for b in range(256):
        s = np.full(64, ord("A"), dtype=np.uint8)
        s[37] = b
        if b in valid:
            w = ctx.encode_array(s)
            assert int(w[1]) == valid[b] << (2 * 5) and int(w[0]) == 0
        else:
            with pytest.raises(bn.NucleotideError) as ei:
                ctx.encode_array(s)
            assert (ei.value.byte, ei.value.index) == (b, 37)


def test_argument_errors(ctx):
    import bitnuc_amd as bn
    with pytest.raises(bn.NucleotideError) as ei:
        ctx.decode_array(np.zeros(1, np.uint64), 33)  # unpacking/mod.rs:40-45
    assert ei.value.kind == "InvalidLength" and ei.value.len == 33
    assert ctx.decode_array(np.zeros(0, np.uint64), 0).size == 0  # n_bases = 0 -> Ok, nothing
    with pytest.raises(bn.NucleotideError) as ei:
        ctx.as_2bit(b"N" * 33)  # length checked before bases
    assert ei.value.kind == "SequenceTooLong" and ei.value.len == 33
    with pytest.raises(bn.NucleotideError) as ei:
        ctx.as_2bit_batch(b"A" * 100, 33)
    assert ei.value.kind == "SequenceTooLong"
    with pytest.raises(RuntimeError):
        ctx.encode_array(b"")  # the reference panics on empty input


# ---- device-pointer path, unaligned pointers, async error latch ---------------------------
def test_dev_path_unaligned_and_async_errors(ctx, oracle):
    import torch
    import bitnuc_amd as bn
    dev = torch.device("cuda:0")
    n = 300007
    s = rand_seq(n)
    for in_off in (0, 1, 3, 8, 13):
        for out_off in (0, 5):
            buf = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
            buf[in_off:in_off + n] = torch.from_numpy(s).to(dev)
            words = torch.zeros((n + 31) // 32 + 2, dtype=torch.int64, device=dev)
            back = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            ctx.encode_dev(buf.data_ptr() + in_off, n, words.data_ptr() + 8)  # 8-byte aligned only
            ctx.decode_dev(words.data_ptr() + 8, (n + 31) // 32, n, back.data_ptr() + out_off)
            ctx.sync()
            w = words.cpu().numpy().view(np.uint64)
            assert w[0] == 0 and w[-1] == 0  # no stray writes around the packed buffer
            assert np.array_equal(w[1:-1], oracle.encode(s))
            b = back.cpu().numpy()
            assert bytes(b[out_off:out_off + n]) == bytes(s).upper()
            assert not b[:out_off].any() and not b[out_off + n:].any()
    # errors are latched per launch and reported in launch order at sync
    good = torch.from_numpy(s).to(dev)
    bad1 = good.clone(); bad1[1234] = ord("N")
    bad2 = good.clone(); bad2[7] = ord("Z")
    words = torch.zeros((n + 31) // 32, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.encode_dev(good, n, words)
    ctx.encode_dev(bad1, n, words)
    ctx.encode_dev(bad2, n, words)
    with pytest.raises(bn.NucleotideError) as ei:
        ctx.sync()
    assert (ei.value.byte, ei.value.index) == (ord("N"), 1234)
    ctx.sync()  # cleared
    # a synchronous host-pointer call between an async error and its sync reports only its
    # own result; the latched async error is kept for the next sync
    ctx.encode_dev(bad2, n, words)
    assert ctx.as_2bit(b"ACGT") == 0xE4
    assert np.array_equal(ctx.encode_array(s[:1000]), oracle.encode(s[:1000]))
    with pytest.raises(bn.NucleotideError) as ei:
        ctx.sync()
    assert (ei.value.byte, ei.value.index) == (ord("Z"), 7)
    ctx.sync()


def test_kmer_dev_paths_unaligned(ctx, sweep_ctx, oracle):
    import torch
    dev = torch.device("cuda:0")
    for k, stride, count in [(31, 31, 5000), (32, 32, 777), (16, 16, 64), (31, 31, 63), (21, 40, 3000)]:
        nbytes = (count - 1) * stride + k
        s = rand_seq(nbytes)
        for off in (0, 1, 7, 16):
            buf = torch.zeros(nbytes + 64, dtype=torch.uint8, device=dev)
            buf[off:off + nbytes] = torch.from_numpy(s).to(dev)
            out = torch.zeros(count + 2, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            ctx.as_2bit_batch_dev(buf.data_ptr() + off, k, stride, count, out.data_ptr() + 8)
            ctx.sync()
            o = out.cpu().numpy().view(np.uint64)
            assert o[0] == 0 and o[-1] == 0
            assert np.array_equal(o[1:-1], oracle.as_2bit_batch(s, k, stride, count)), (k, stride, count, off)
    # dense-kernel switch off == on
    s = rand_seq(31 * 4096)
    assert ctx.set_variant("batch_dense", 0) == -2  # a switch of the evidence build: the product holds the shipped routing only
    prev = sweep_ctx.set_variant("batch_dense", 0)
    a = sweep_ctx.as_2bit_batch(s, 31, 31, 4096)
    sweep_ctx.set_variant("batch_dense", 1)
    b = sweep_ctx.as_2bit_batch(s, 31, 31, 4096)
    sweep_ctx.set_variant("batch_dense", prev)
    assert np.array_equal(a, b) and np.array_equal(a, oracle.as_2bit_batch(s, 31, 31, 4096)) and np.array_equal(a, ctx.as_2bit_batch(s, 31, 31, 4096))
    # scan with unaligned ref / dist pointers
    n, k = 70001, 31
    s = rand_seq(n)
    q = oracle.as_2bit(rand_seq(k))
    for off, doff in [(0, 0), (3, 0), (0, 5), (9, 2)]:
        buf = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
        buf[off:off + n] = torch.from_numpy(s).to(dev)
        d = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ctx.kmer_hdist_scan_dev(buf.data_ptr() + off, n, k, q, d.data_ptr() + doff)
        ctx.sync()
        h = d.cpu().numpy()
        assert np.array_equal(h[doff:doff + n - k + 1], oracle.kmer_hdist_scan(s, k, q)), (off, doff)
        assert not h[:doff].any() and not h[doff + n - k + 1:].any()
    # dense batch: invalid byte found in the whole-wave part and in the leftover part
    import bitnuc_amd as bn
    for pos in (5, 31 * 64 * 3 + 17, 31 * 4100 + 2):
        t = rand_seq(31 * 4130).copy()
        t[pos] = ord("N")
        with pytest.raises(bn.NucleotideError) as ei:
            ctx.as_2bit_batch(t, 31, 31, 4130)
        assert (ei.value.byte, ei.value.index) == (ord("N"), pos)


def test_nucgen_matches_host_generator(ctx, oracle):
    import torch
    dev = torch.device("cuda:0")
    for n, first, flags in [(1000, 0, 0), (100003, 32 * 77, 0), (4097, 5, 0), (333, 1 << 40, 0), (1000, 3, 1), (70, 0, 1)]:
        t = torch.zeros(n + 16, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        ctx.nucgen_dev(t, n, 0xB17C0DE, first, flags)
        ctx.sync()
        h = t.cpu().numpy()
        assert np.array_equal(h[:n], oracle.nucgen(n, 0xB17C0DE, first, flags)), (n, first, flags)
        assert not h[n:].any()


# ---- k-mer batch and scan ---------------------------------------------------------------------
@pytest.mark.parametrize("k,stride", [(31, 31), (31, 32), (32, 32), (1, 1), (4, 1), (21, 21), (31, 1), (16, 50),
                                      (31, 64), (31, 65), (27, 200), (32, 33), (3, 7)])
def test_kmer_batch_vs_oracle(ctx, oracle, k, stride):
    for count in [1, 2, 255, 256, 257, 10007]:
        s = rand_seq((count - 1) * stride + k)
        got = ctx.as_2bit_batch(s, k, stride, count)
        assert np.array_equal(got, oracle.as_2bit_batch(s, k, stride, count)), (k, stride, count)


@pytest.mark.parametrize("k", [1, 2, 4, 15, 16, 17, 21, 31, 32])
def test_kmer_windows_stride1_vs_oracle(ctx, sweep_ctx, oracle, k):
    """`for w in seq.windows(k) { as_2bit(w) }` (src/lib.rs:170-173): the sliding kernel (whole 1 KiB rounds) plus
    the generic kernel for the leftover windows, against the oracle's loop; sizes around the round boundaries."""
    import torch
    dev = torch.device("cuda:0")
    for n in [k, 1023, 1024, 1025, 1024 + 991, 1024 + 992, 1024 + 993, 3000, 5 * 992 + 1024, 200003]:
        if n < k:
            continue
        s = rand_seq(n)
        count = n - k + 1
        exp = oracle.as_2bit_batch(s, k, 1, count)
        assert np.array_equal(ctx.as_2bit_batch(s, k, 1, count), exp), (k, n)
    # device path at 16-byte and odd alignment (the latter takes the generic kernel), guard words around the output
    n = 100003
    s = rand_seq(n)
    count = n - k + 1
    exp = oracle.as_2bit_batch(s, k, 1, count)
    for in_off, out_off in ((0, 0), (16, 2), (5, 0), (0, 1)):
        buf = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
        buf[in_off:in_off + n] = torch.from_numpy(s).to(dev)
        out = torch.zeros(count + 4, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        ctx.as_2bit_batch_dev(buf.data_ptr() + in_off, k, 1, count, out.data_ptr() + 8 * out_off)
        ctx.sync()
        o = out.cpu().numpy().view(np.uint64)
        assert np.array_equal(o[out_off:out_off + count], exp), (k, in_off, out_off)
        assert not o[:out_off].any() and not o[out_off + count:].any()
    # sliding kernel off == on
    prev = sweep_ctx.set_variant("batch_slide", 0)  # (evidence build: the product has no such switch)
    try:
        assert np.array_equal(sweep_ctx.as_2bit_batch(s, k, 1, count), exp)
    finally:
        sweep_ctx.set_variant("batch_slide", prev)


@pytest.mark.parametrize("stride", [2, 4, 8, 16])
@pytest.mark.parametrize("k", [1, 7, 8, 16, 17, 31, 32])
def test_kmer_windows_small_power_of_two_strides(ctx, oracle, k, stride):
    import bitnuc_amd as bn
    for count in [1, 63, 64, 65, 991 // stride, 1024 // stride + 1, 2016 // stride + 3, 10007, 100003]:
        n = (count - 1) * stride + k
        s = rand_seq(n)
        assert np.array_equal(ctx.as_2bit_batch(s, k, stride, count), oracle.as_2bit_batch(s, k, stride, count)), (k, stride, count)
    # error position and gaps: with k < stride the bytes between k-mers are never examined
    count = 5000
    n = (count - 1) * stride + k
    s = rand_seq(n).copy()
    if k < stride:
        s[stride * 100 + k] = ord("N")  # first byte of a gap
        assert np.array_equal(ctx.as_2bit_batch(s, k, stride, count), oracle.as_2bit_batch(s, k, stride, count))
    pos = stride * 3000 + min(k, stride) - 1
    s[pos] = ord("N")
    s[pos + stride] = ord("X")
    with pytest.raises(bn.NucleotideError) as ei:
        ctx.as_2bit_batch(s, k, stride, count)
    with pytest.raises(oracle.OracleError) as oe:
        oracle.as_2bit_batch(s, k, stride, count)
    assert (ei.value.byte, ei.value.index) == (oe.value.byte, oe.value.index) == (ord("N"), pos)


@pytest.mark.parametrize("k,stride", [(31, 3), (31, 5), (31, 6), (31, 7), (21, 12), (31, 24), (32, 31), (32, 17), (16, 9), (7, 3), (31, 30)])
def test_kmer_windows_other_small_strides(ctx, sweep_ctx, oracle, k, stride):
    """Overlapping k-mers at a stride that is not a power of two: the sliding round with per-lane window selection."""
    import bitnuc_amd as bn
    for count in [1, 40, 330 // max(1, stride // 3), 1024 // stride + 1, 2016 // stride + 3, 5 * 992 // stride + 7, 10007, 100003]:
        n = (count - 1) * stride + k
        s = rand_seq(n)
        assert np.array_equal(ctx.as_2bit_batch(s, k, stride, count), oracle.as_2bit_batch(s, k, stride, count)), (k, stride, count)
    count = 6000
    n = (count - 1) * stride + k
    s = rand_seq(n).copy()
    pos = stride * 3777 + k - 1
    s[pos] = ord("N")
    s[min(n - 1, pos + 2 * stride)] = ord("X")
    with pytest.raises(bn.NucleotideError) as ei:
        ctx.as_2bit_batch(s, k, stride, count)
    with pytest.raises(oracle.OracleError) as oe:
        oracle.as_2bit_batch(s, k, stride, count)
    assert (ei.value.byte, ei.value.index) == (oe.value.byte, oe.value.index) == (ord("N"), pos)
    prev = sweep_ctx.set_variant("batch_slide", 0)  # general kernel == sliding kernel (evidence build: the product has no such switch)
    try:
        t = rand_seq((20000 - 1) * stride + k)
        ref = sweep_ctx.as_2bit_batch(t, k, stride, 20000)
    finally:
        sweep_ctx.set_variant("batch_slide", prev)
    assert np.array_equal(ctx.as_2bit_batch(t, k, stride, 20000), ref)


def test_kmer_windows_stride1_first_invalid_byte(ctx, oracle):
    import bitnuc_amd as bn
    k, n = 31, 50000
    s = rand_seq(n).copy()
    for pos in (0, 15, 16, 991, 992, 1023, 1024, 20000, n - 1):
        t = s.copy()
        t[pos] = ord("N")
        if pos + 7 < n:
            t[pos + 7] = ord("X")  # a later invalid byte never wins
        with pytest.raises(bn.NucleotideError) as ei:
            ctx.as_2bit_batch(t, k, 1, n - k + 1)
        with pytest.raises(oracle.OracleError) as oe:
            oracle.as_2bit_batch(t, k, 1, n - k + 1)
        assert (ei.value.byte, ei.value.index) == (oe.value.byte, oe.value.index) == (ord("N"), pos)
    # a byte past the last window is never examined
    t = np.concatenate([s, np.frombuffer(b"N", dtype=np.uint8)])
    assert np.array_equal(ctx.as_2bit_batch(t, k, 1, n - k + 1), oracle.as_2bit_batch(s, k, 1, n - k + 1))


def test_kmer_batch_errors(ctx, oracle):
    import bitnuc_amd as bn
    k, stride, count = 31, 40, 3000
    s = rand_seq((count - 1) * stride + k).copy()
    s[stride * 100 + 35] = ord("N")   # in a gap between k-mers: not examined
    assert np.array_equal(ctx.as_2bit_batch(s, k, stride, count), oracle.as_2bit_batch(s, k, stride, count))
    s[stride * 2000 + 30] = ord("N")
    s[stride * 1500 + 3] = ord("Q")
    with pytest.raises(bn.NucleotideError) as ei:
        ctx.as_2bit_batch(s, k, stride, count)
    with pytest.raises(oracle.OracleError) as oe:
        oracle.as_2bit_batch(s, k, stride, count)
    assert (ei.value.byte, ei.value.index) == (oe.value.byte, oe.value.index) == (ord("Q"), stride * 1500 + 3)
    assert np.array_equal(ctx.as_2bit_batch(b"", 0, 1, 5), np.zeros(5, np.uint64))


# Forms of the config-5 scan.  "ships" runs on the PRODUCT library: the one-hot contraction on the matrix cores in the fused count's tiling (scan_impl 8: segments
# of 32 windows x 32 shifts, four MFMAs per 1024 windows, two v_permlane32_swap put the packed results in store order, one trip of 4 rounds per wave, workgroups of one wave).  The others
# live in the evidence build: its trips of 2 / 3 rounds and workgroups of 2 / 4 waves; the same tiling with three channels per base (scan_mfma_ch3: three MFMAs, 1.5 % slower); the natural-layout tiling that shipped first (scan_impl 7: six MFMAs, results already in store order) with
# its operand / pack / trip / grid forms (shift: 0 global re-loads, 1 bytes through the strip, 2 DPP, 3 all six operands through the strip, 4 the lane's own kept in
# registers, 5 the software-pipelined trip, 6 less bookkeeping; pack: 0 v_cvt_pk_u8, 2 bias by a seventh instruction; match: the table marks the equal channel and counts down from k)
# and rounds 1-4's bit-plane forms (scan_impl, scan_unroll): 1 = line-aligned rounds of 1024 windows (GEN 1 at unroll 4: shipped in round 4), 6 = rounds 2-3's plane
# build, 0 = rounds of 992 windows, 2 / 3 / 4 = kmer_scan3_kernel with 12 / 20 / 16 rounds per wave.
SCAN_FORMS = [("ships", {})] + \
    [(f"seg-U{u}", dict(scan_impl=8, scan_mfma_unroll=u)) for u in (3, 2)] + [(f"seg-workgroups-of-{b}", dict(scan_impl=8, scan_mfma_block=b)) for b in (128, 256)] + [(f"seg-three-channels-U{u}", dict(scan_impl=8, scan_mfma_unroll=u, scan_mfma_ch3=1)) for u in (4, 3, 2)] + \
    [(f"mfma-shift{sh}-pack{pk}-U{u}-persist{ps}" + ("-match" if mt else ""), dict(scan_impl=7, scan_mfma_shift=sh, scan_mfma_pack=pk, scan_mfma_unroll=u, scan_mfma_persist=ps, scan_mfma_match=mt))
     for sh, pk, u, ps, mt in ((4, 1, 4, 0, 0), (4, 1, 4, 1, 0), (4, 0, 2, 0, 0), (4, 2, 4, 0, 0), (4, 1, 3, 0, 0), (5, 0, 4, 0, 0), (5, 1, 2, 1, 0), (3, 1, 4, 0, 0), (3, 0, 2, 1, 0), (1, 1, 4, 0, 0), (1, 2, 2, 1, 0), (2, 1, 4, 0, 0), (2, 0, 2, 1, 0),
                               (0, 1, 2, 1, 0), (0, 0, 2, 0, 0), (6, 1, 4, 0, 0), (4, 1, 4, 0, 1), (6, 1, 2, 1, 1))] + \
    [(f"bitplane-impl{i}-unroll{u}", dict(scan_impl=i, scan_unroll=u)) for i, u in ((1, 4), (1, 2), (1, 1), (6, 4), (0, 4), (0, 2), (0, 1), (2, 4), (3, 4), (4, 4))]
SCAN_DEFAULTS = dict(scan_impl=8, scan_mfma_block=64, scan_mfma_ch3=0, scan_mfma_match=0, scan_unroll=4, scan_mfma_shift=4, scan_mfma_pack=1, scan_mfma_unroll=4, scan_mfma_persist=0, scan_mfma_count_persist=1, scan_mfma_count_form=2, scan_mfma_count_emit=2, scan_mfma_count_rounds=4, scan_mfma_count_grid=12)


@pytest.mark.parametrize("form", SCAN_FORMS, ids=[name for name, _ in SCAN_FORMS])
@pytest.mark.parametrize("k", [1, 2, 15, 16, 17, 31, 32])
def test_scan_vs_oracle(ctx, sweep_ctx, oracle, k, form):
    name, knobs = form
    if knobs:
        ctx = sweep_ctx  # the product ships only the form in use; the alternatives live in the evidence build
        for key, v in {**SCAN_DEFAULTS, **knobs}.items():
            ctx.require_variant(key, v)
    else:
        assert ctx.get("sweep_build") == 0 and ctx.get("scan_impl") == 8
    try:
        # (12 / 16 / 20 rounds per wave: sizes around one and two chunks as well; 4 rounds per trip: around 4 KiB + the 32-byte halo)
        for n in [k, k + 1, 1000, 1023, 1024, 1025, 1055, 1056, 1057, 2015, 2016, 2017, 2047, 2048, 2079, 2080, 2081, 3103, 3104, 3105, 4127, 4128, 4129, 5000, 5152, 5153,
                  12 * 1024 + 31, 12 * 1024 + 32, 12 * 1024 + 33, 16 * 1024 + 31, 16 * 1024 + 32, 16 * 1024 + 33, 17 * 1024 + 32, 20 * 1024 + 32, 20 * 1024 + 33, 24 * 1024 + 33, 32 * 1024 + 31, 32 * 1024 + 32, 32 * 1024 + 33, 33 * 1024 + 40, 64 * 1024 + 32, 65 * 1024 + 100, 200003]:
            s = rand_seq(n)
            q = int(RNG.integers(0, 1 << 62)) | (int(RNG.integers(0, 4)) << 62)
            got = ctx.kmer_hdist_scan(s, k, q)
            assert np.array_equal(got, oracle.kmer_hdist_scan(s, k, q)), (k, n)
    finally:
        if knobs:
            for key, v in SCAN_DEFAULTS.items():
                ctx.require_variant(key, v)


MATRIX_FORMS = [f for f in SCAN_FORMS if f[0] == "ships" or f[0].startswith(("mfma", "seg"))]


@pytest.mark.parametrize("form", MATRIX_FORMS, ids=[name for name, _ in MATRIX_FORMS])
def test_scan_matrix_core_forms_first_invalid_byte_and_count(ctx, sweep_ctx, oracle, form):
    """kmer_scan_seg_mfma_kernel / kmer_scan_mfma_kernel: the first invalid byte wins at round, trip and strip boundaries and inside the halo (a later invalid byte never
    does, a byte after the last window is never examined: hamming/scalar.rs:11-48 over naive.rs:3-20 per window); bytes past the last window are
    not written; the fused count (bitnuc_kmer_hdist_count_dev) of every form equals the count over the oracle's distance bytes."""
    import bitnuc_amd as bn
    import torch
    name, knobs = form
    if knobs:
        ctx = sweep_ctx
        for key, v in {**SCAN_DEFAULTS, **knobs}.items():
            ctx.require_variant(key, v)
        if knobs["scan_impl"] == 7:
            ctx.require_variant("scan_mfma_count_form", 0)  # the fused count on the natural-layout tiling; "ships", "seg-*" and test_scan_fused_threshold_count run its own tiling
            ctx.require_variant("scan_mfma_count_persist", knobs["scan_mfma_persist"] ^ 1)  # ... in its two grid forms (resident + ticket / one trip per wave + finishing launch)
    try:
        rng = np.random.default_rng(505)
        n = 9 * 1024 + 77
        s = ALPHA8[rng.integers(0, 8, size=n)].copy()
        k, q = 31, 0x0123456789ABCDEF & ((1 << 62) - 1)
        for pos in (0, 15, 16, 1023, 1024, 1025, 1039, 1040, 1055, 1056, 2047, 2048, 4095, 4096, 4097, 4 * 1024 + 31, 4 * 1024 + 32, 8 * 1024 - 1, 8 * 1024, 8 * 1024 + 33, n - k - 1, n - 1):
            t = s.copy()
            t[pos] = ord("N")
            if pos + 9 < n:
                t[pos + 9] = ord("X")  # a later invalid byte never wins
            with pytest.raises(bn.NucleotideError) as ei:
                ctx.kmer_hdist_scan(t, k, q)
            assert (ei.value.byte, ei.value.index) == (ord("N"), pos), pos
        dev = torch.device("cuda:0")
        for m in (k, 1056, 1057, 4128, 4129, n):
            tt = torch.from_numpy(s[:m]).to(dev)
            d = torch.full((m + 64,), 0xEE, dtype=torch.uint8, device=dev)
            cnt = torch.zeros(1, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            ctx.kmer_hdist_scan_dev(tt, m, k, q, d)
            ctx.kmer_hdist_count_dev(tt, m, k, q, 22, cnt)
            ctx.sync()
            want = oracle.kmer_hdist_scan(s[:m], k, q)
            got = d.cpu().numpy()
            assert np.array_equal(got[:m - k + 1], want) and bool((got[m - k + 1:] == 0xEE).all()), m
            assert int(cnt.item()) == int((want <= 22).sum()), m
    finally:
        if knobs:
            for key, v in SCAN_DEFAULTS.items():
                ctx.require_variant(key, v)


@pytest.mark.parametrize("form,U,grid,emit", [(2, 4, 12, 2), (2, 3, 18, 2), (2, 2, 24, 2), (2, 4, 1, 2), (1, 3, 18, 2), (1, 3, 18, 0), (1, 3, 18, 1), (1, 4, 4, 2), (1, 2, 16, 1), (1, 4, 12, 0), (1, 3, 1, 2)],
                         ids=["ships", "evidence-three-channels-trips-of-3", "evidence-three-channels-trips-of-2", "evidence-three-channels-one-workgroup-per-CU",
                              "evidence-four-channels", "evidence-four-channels-compare-per-register", "evidence-four-channels-second-register-set", "evidence-four-channels-trips-of-4",
                              "evidence-four-channels-trips-of-2", "evidence-four-channels-trips-of-4-compare-per-register", "evidence-four-channels-one-workgroup-per-CU"])
def test_fused_count_own_tiling_vs_oracle(ctx, sweep_ctx, oracle, form, U, grid, emit):
    """kmer_count3_mfma_kernel (segments of 32 windows x 32 shifts, three channels per base: 3 MFMAs per 1024 windows, the threshold inside the product) and
    kmer_count_mfma_kernel (four channels: 4 MFMAs; the threshold inside the product or compared per register): the
    count of d <= tau for every k, at sizes around the rounds, the trips and the 32-byte halo, on random data and on data where most windows are hits, for
    thresholds on both sides of k (tau >= k: every window counts; tau up to 2^32 - 1), equals the count over the oracle's distance bytes
    (hamming/scalar.rs:11-48 over naive.rs:3-20 per window); the first invalid byte is reported with its index; back-to-back launches find the
    accumulator re-armed."""
    import bitnuc_amd as bn
    import torch
    dev = torch.device("cuda:0")
    shipped = (form, U, grid, emit) == (2, 4, 12, 2)
    if not shipped:
        ctx = sweep_ctx
        for key, v in {**SCAN_DEFAULTS, "scan_mfma_count_form": form, "scan_mfma_count_rounds": U, "scan_mfma_count_grid": grid, "scan_mfma_count_emit": emit}.items():
            ctx.require_variant(key, v)
    try:
        assert ctx.get("scan_mfma_count_form") == form and ctx.get("scan_mfma_count_rounds") == U and ctx.get("scan_mfma_count_emit") == emit
        rng = np.random.default_rng(41 + U + 7 * form)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        for k in (1, 2, 15, 16, 17, 31, 32):
            for n in (k, 1055, 1056, 1057, 2080, 2081, 3104, 3105, 4128, 4129, 5152, 5153, 6 * 1024 + 32, 6 * 1024 + 33, 8 * 1024 + 32, 9 * 1024 + 77, 200003, 3 * 10**6 + 77):
                if n < k or (n > 10**6 and k not in (31, 32)):
                    continue
                q = int(rng.integers(0, 1 << 62)) | (int(rng.integers(0, 4)) << 62)
                for kind in ("random", "periodic"):
                    if kind == "random":
                        s = ALPHA8[rng.integers(0, 8, size=n)]
                    elif n > 10**6:
                        continue
                    else:  # the query's own bases repeated, with a few substitutions: the windows at multiples of k are hits or near hits
                        unit = np.array([ord("ACGT"[(q >> (2 * i)) & 3]) for i in range(k)], dtype=np.uint8)
                        s = np.tile(unit, n // k + 1)[:n].copy()
                        s[rng.integers(0, n, size=max(1, n // 50))] = ord("a")
                    t = torch.from_numpy(s).to(dev)
                    d = oracle.kmer_hdist_scan(s, k, q)
                    for tau in sorted({0, 1, k // 2, k - 1, k, k + 1, 31, 32, 33, 2**32 - 1}):
                        torch.cuda.synchronize()
                        ctx.kmer_hdist_count_dev(t, n, k, q, tau, cnt)
                        ctx.kmer_hdist_count_dev(t, n, k, q, tau, cnt)
                        ctx.sync()
                        assert int(cnt.item()) == int((d <= tau).sum()), (k, n, tau, kind)
        s = ALPHA8[rng.integers(0, 8, size=50000)].copy()
        for pos in (0, 15, 16, 1023, 1024, 1040, 1055, 1056, 3071, 3072, 3104, 4095, 4096, 4097, 4127, 4128, 30000, 49999):
            b = s.copy()
            b[pos] = ord("N")
            if pos + 9 < b.size:
                b[pos + 9] = ord("X")
            tb = torch.from_numpy(b).to(dev)
            torch.cuda.synchronize()
            ctx.kmer_hdist_count_dev(tb, b.size, 31, 0, 3, cnt)
            with pytest.raises(bn.NucleotideError) as ei:
                ctx.sync()
            assert (ei.value.byte, ei.value.index) == (ord("N"), pos), pos
    finally:
        if not shipped:
            for key, v in SCAN_DEFAULTS.items():
                ctx.require_variant(key, v)


def test_scan_fused_threshold_count(ctx, oracle):
    """bitnuc_kmer_hdist_count_dev: the number of windows with d <= tau equals the count over the distance bytes of the
    plain scan (oracle), at sizes around the 1024-window rounds and their 32-byte halo, for several k and tau."""
    import torch
    dev = torch.device("cuda:0")
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    for n in (31, 40, 1055, 1056, 1057, 2079, 2080, 2081, 4 * 1024 + 32, 16 * 1024 + 31, 100003, 1 << 20):
        s = rand_seq(n)
        t = torch.from_numpy(s).to(dev)
        for k, tau in ((31, 20), (31, 0), (32, 24), (16, 9), (1, 0), (7, 7)):
            if n < k:
                continue
            q = int(RNG.integers(0, 1 << 62)) & ((1 << (2 * k)) - 1)
            if tau == 0 and n > 2000:  # plant exact matches so that the tau = 0 count is not trivially zero
                q = oracle.as_2bit(s[1500:1500 + k])
            torch.cuda.synchronize()
            ctx.kmer_hdist_count_dev(t, n, k, q, tau, cnt)
            ctx.sync()
            d = oracle.kmer_hdist_scan(s, k, q)
            assert int(cnt.item()) == int((d <= tau).sum()), (n, k, tau)
    # back-to-back launches reuse the context's accumulator (it must be zero again after each launch)
    for _ in range(3):
        ctx.kmer_hdist_count_dev(t, n, k, q, tau, cnt)
    ctx.sync()
    assert int(cnt.item()) == int((d <= tau).sum())
    # unaligned reference pointer, no windows, invalid base
    buf = torch.from_numpy(np.concatenate([np.zeros(3, np.uint8), s])).to(dev)
    torch.cuda.synchronize()
    ctx.kmer_hdist_count_dev(buf.data_ptr() + 3, n, k, q, tau, cnt)
    ctx.sync()
    assert int(cnt.item()) == int((d <= tau).sum())
    ctx.kmer_hdist_count_dev(t, 5, 31, 0, 3, cnt)
    ctx.sync()
    assert int(cnt.item()) == 0
    import bitnuc_amd as bn
    bad = s.copy()
    bad[70000] = ord("N")
    tb = torch.from_numpy(bad).to(dev)
    torch.cuda.synchronize()
    ctx.kmer_hdist_count_dev(tb, n, 31, 0, 3, cnt)
    with pytest.raises(bn.NucleotideError) as ei:
        ctx.sync()
    assert (ei.value.byte, ei.value.index) == (ord("N"), 70000)


def test_scan_errors_and_bench_invariant(ctx, oracle):
    import bitnuc_amd as bn
    s = rand_seq(50000).copy()
    for pos in (0, 991, 992, 1007, 1008, 1023, 1024, 1040, 1055, 30000, 49600, 49999):
        t = s.copy()
        t[pos] = ord("N")
        with pytest.raises(bn.NucleotideError) as ei:
            ctx.kmer_hdist_scan(t, 31, 0)
        assert (ei.value.byte, ei.value.index) == (ord("N"), pos)
    assert ctx.kmer_hdist_scan(s[:10], 31, 0).size == 0
    with pytest.raises(bn.NucleotideError):
        ctx.kmer_hdist_scan(s, 33, 0)
    # hdist_benchmark.rs:17-37 shape: cyc-4 reference vs the cyc-3 32-mer as query
    ref = np.array([ALPHA[i % 4] for i in range(4096)], dtype=np.uint8)
    qs = np.array([ALPHA[i % 3] for i in range(32)], dtype=np.uint8)
    d = ctx.kmer_hdist_scan(ref, 32, ctx.as_2bit(qs))
    for i in range(8):
        assert d[i] == int((ref[i:i + 32] != qs).sum())


# ---- ragged batches of independent sequences ----------------------------------------------------
def _ragged(lengths, alpha=ALPHA8):
    off = np.zeros(len(lengths) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lengths)
    return rand_seq(int(off[-1]), alpha), off


def _oracle_batch(oracle, seq, off):
    words, wo = [], [0]
    for i in range(len(off) - 1):
        s = seq[int(off[i]):int(off[i + 1])]
        w = oracle.encode(s) if len(s) else np.zeros(0, np.uint64)  # host loop of encode(), one call per sequence
        words.append(w)
        wo.append(wo[-1] + len(w))
    return (np.concatenate(words) if words else np.zeros(0, np.uint64)), np.array(wo, dtype=np.uint64)


@pytest.fixture(params=[(1, 1, 1), (1, 2, 1), (1, 4, 1), (0, 1, 1), (0, 1, 0)], ids=["plan", "plan-2tiles", "plan-4tiles", "tables", "tables-r2-kernels"])
def batch_ctx(request, ctx, sweep_ctx):
    """A context set to one formulation of the ragged-batch kernels behind the host-pointer entry points: the layout plan
    (bitnuc_batch_plan: one pad byte per word; what host calls use), the table-driven entry points (round 3: one asynchronous
    pass emits the plan from the two offset tables into context scratch, then the plan kernels), or round 2's table-driven
    kernels (tile records by a search pre-kernel + O(1) pad-scatter lookup), which live on in the evidence build.  The plan
    kernels with 2 and 4 tiles per wave trip exist in the evidence build only (they lost their A/B)."""
    use_plan, tiles, impl = request.param
    c = ctx if (use_plan, tiles, impl) == (1, 1, 1) else sweep_ctx  # the product holds the shipped routing only
    prev = c.set_variant("batch_host_plan", use_plan)
    prev_e = c.set_variant("plan_enc_tiles", tiles)
    prev_d = c.set_variant("plan_tiles", tiles)
    prev_i = c.set_variant("batch_tables_impl", impl)
    assert c.get("plan_enc_tiles") == tiles and c.get("plan_tiles") == tiles and c.get("batch_tables_impl") == impl
    yield c
    c.set_variant("batch_host_plan", prev)
    c.set_variant("plan_enc_tiles", prev_e)
    c.set_variant("plan_tiles", prev_d)
    c.set_variant("batch_tables_impl", prev_i)


@pytest.mark.parametrize("shape", ["reads150", "tiny", "mixed", "with_empties", "one_long", "many_empties", "len32", "ones_and_empties", "unaligned_long"])
def test_batch_encode_decode_vs_oracle_loop(batch_ctx, oracle, shape):
    ctx = batch_ctx
    lengths = {
        "reads150": [150] * 3000,
        "tiny": list(RNG.integers(1, 5, size=5000)),
        "mixed": list(RNG.integers(1, 400, size=2000)) + [100000, 31, 32, 33, 64, 1],
        "with_empties": [0, 0, 5, 0, 37, 0, 0, 0, 64, 0] * 200,
        "one_long": [1000003],
        "many_empties": [0] * 1000 + [40] + [0] * 2000 + [7, 0, 0, 33] + [0] * 500,
        "len32": [32] * 4099,                                   # 64 sequence starts per tile: the second window round
        "ones_and_empties": [1, 0, 1, 1, 0, 0, 1] * 700 + [31, 1, 33, 0, 1] * 50,  # > 64 starts per tile incl. empties
        "unaligned_long": [7, 300001, 13, 2049, 2048, 2047, 5],  # dense tiles whose first base is not 16-byte aligned (chunk 128)
    }[shape]
    seq, off = _ragged(lengths)
    ew, ewo = _oracle_batch(oracle, seq, off)
    w, wo = ctx.encode_batch(seq, off)
    assert np.array_equal(wo, ewo), shape
    assert np.array_equal(w, ew), shape
    back = ctx.decode_batch(w, wo, off)
    assert bytes(back) == bytes(seq).upper(), shape


def test_batch_offsets_base_and_errors(batch_ctx, oracle):
    ctx = batch_ctx
    import bitnuc_amd as bn
    lengths = list(RNG.integers(1, 300, size=1500))
    seq, off = _ragged(lengths)
    # a batch that starts in the middle of the buffer (offsets[0] != 0), unaligned
    pre = 37
    buf = np.concatenate([np.full(pre, ord("N"), np.uint8), seq, np.full(11, ord("N"), np.uint8)])
    off2 = off + np.uint64(pre)
    w, wo = ctx.encode_batch(buf, off2)
    ew, ewo = _oracle_batch(oracle, seq, off)
    assert np.array_equal(w, ew) and np.array_equal(wo, ewo)
    back = ctx.decode_batch(w, wo, off2)
    assert bytes(back[pre:pre + len(seq)]) == bytes(seq).upper() and not back[:pre].any()
    # first invalid byte in buffer order, reported with its byte offset
    bad = seq.copy()
    p1, p2 = int(off[700]) + 3, int(off[900])
    bad[p1], bad[p2] = ord("N"), ord("X")
    with pytest.raises(bn.NucleotideError) as ei:
        ctx.encode_batch(bad, off)
    assert (ei.value.kind, ei.value.byte, ei.value.index) == ("InvalidBase", ord("N"), p1)
    # decreasing offsets
    off_bad = off.copy()
    off_bad[10] = off_bad[9] - np.uint64(1)
    with pytest.raises(bn.NucleotideError) as ei:
        ctx.encode_batch(seq, off_bad)
    assert ei.value.kind == "InvalidRange"
    # a word-offsets table that does not belong to these offsets is refused, not dereferenced
    w, wo = ctx.encode_batch(seq, off)
    wo_bad = wo.copy()
    wo_bad[20:] += np.uint64(3)
    with pytest.raises(bn.NucleotideError) as ei:
        ctx.decode_batch(np.concatenate([w, np.zeros(3, np.uint64)]), wo_bad, off)
    assert ei.value.kind == "InvalidRange" and ei.value.index == 20
    with pytest.raises(ValueError):
        ctx.decode_batch(w[:-1], wo, off)
    with pytest.raises(ValueError):
        ctx.encode_batch(seq[:-1], off)
    with pytest.raises(ValueError):
        ctx.as_2bit_batch(seq[:100], 31, 31, 4)
    with pytest.raises(ValueError):
        ctx.encode_fixed(seq[:100], 30, 30, 4)
    # empty batch
    w, wo = ctx.encode_batch(b"", np.zeros(1, np.uint64))
    assert w.size == 0 and list(wo) == [0]


def test_batch_full_scale_reads(ctx, oracle):
    """~6.7 M reads of 150 bases (10^9 bases): device path, spot-checked against the oracle loop."""
    import torch
    dev = torch.device("cuda:0")
    L, count = 150, 6_666_666
    n = L * count
    seq = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.nucgen_dev(seq, n, 0xB17C0DE)
    off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L
    wo = torch.empty(count + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    total = ctx.batch_word_offsets_dev(off, count, wo)
    assert total == count * 5
    assert torch.equal(wo, torch.arange(0, count + 1, dtype=torch.int64, device=dev) * 5)
    words = torch.empty(total, dtype=torch.int64, device=dev)
    back = torch.zeros(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.encode_batch_dev(seq, off, wo, count, total, words)
    ctx.decode_batch_dev(words, wo, off, count, total, back)
    ctx.sync()
    assert torch.equal(seq, back)
    # the same batch through a layout plan: same word offsets, same words, same bases
    import bitnuc_amd as bn
    plan = bn.BatchPlan(ctx, off, count)
    assert plan.total_words == total
    words_p = torch.zeros(total, dtype=torch.int64, device=dev)
    back.zero_()
    torch.cuda.synchronize()
    plan.encode_dev(seq, words_p)
    plan.decode_dev(words_p, back)
    ctx.sync()
    assert torch.equal(words_p, words) and torch.equal(seq, back)
    plan.close()
    for r0 in (0, 1_234_567, count - 2000):
        h = seq[r0 * L:(r0 + 2000) * L].cpu().numpy()
        exp = np.concatenate([oracle.encode(h[i * L:(i + 1) * L]) for i in range(2000)])
        assert np.array_equal(words[r0 * 5:(r0 + 2000) * 5].cpu().numpy().view(np.uint64), exp)


def test_hdist_bulk_vs_oracle(ctx, oracle):
    for n in [1, 31, 32, 33, 127, 128, 129, 100000, 1000003]:
        a, b = rand_seq(n, ALPHA), rand_seq(n, ALPHA)
        wa, wb = oracle.encode(a), oracle.encode(b)
        assert ctx.hdist(wa, wb, n) == oracle.hdist(wa, wb, n) == int((a != b).sum())
    # bits above n_bases in the last word are ignored (scalar.rs:26-33)
    wa = np.array([0xFFFFFFFFFFFFFFFF], dtype=np.uint64)
    wb = np.array([0], dtype=np.uint64)
    for n in range(0, 33):
        assert ctx.hdist(wa, wb, n) == n == oracle.hdist(wa, wb, n)


# ---- BASELINE-size properties (config 2: 10^9 bases on one GPU) ---------------------------
def test_full_size_roundtrip_and_checksums(ctx, oracle):
    import torch
    dev = torch.device("cuda:0")
    n = 10**9 + 17  # tailed on purpose; BASELINE config 2 is 10^9
    seq = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.nucgen_dev(seq, n, 0xB17C0DE)
    nw = (n + 31) // 32
    words = torch.empty(nw, dtype=torch.int64, device=dev)
    back = torch.empty(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.encode_dev(seq, n, words)
    ctx.decode_dev(words, nw, n, back)
    ctx.sync()
    assert torch.equal(seq, back)  # encode -> decode is the identity on uppercase input
    # generator words ARE the packed words (base i = 2-bit field i of mix(seed, i/32)):
    # an independent closed form for the whole 250 MB output, checked by 64-bit sums per block
    idx = torch.arange(1, nw + 1, dtype=torch.int64, device=dev)
    z = idx * (-7046029254386353131) + 0xB17C0DE  # 0x9E3779B97F4A7C15 as i64, wraps mod 2^64

    def lsr(x, s):  # logical shift right on int64
        return (x >> s) & ((1 << (64 - s)) - 1)
    z = (z ^ lsr(z, 30)) * (-4658895280553007687)   # 0xBF58476D1CE4E5B9
    z = (z ^ lsr(z, 27)) * (-7723592293110705685)   # 0x94D049BB133111EB
    z = z ^ lsr(z, 31)
    z[-1] &= (1 << (2 * (n % 32))) - 1              # last word: 17 bases, zero-padded high
    assert torch.equal(words, z)
    # spot-check blocks against the CPU oracle
    for off in (0, 32 * 1_000_000, (n // 32 - 4096) * 32):
        m = min(32 * 4096, n - off)
        h = seq[off:off + m].cpu().numpy()
        assert np.array_equal(words[off // 32: off // 32 + (m + 31) // 32].cpu().numpy().view(np.uint64), oracle.encode(h))
    # linearity-style property: encoding two halves separately == encoding the whole
    half = (n // 64) * 32
    w2 = torch.empty(nw, dtype=torch.int64, device=dev)
    ctx.encode_dev(seq, half, w2)
    ctx.encode_dev(seq.data_ptr() + half, n - half, w2.data_ptr() + half // 4)
    ctx.sync()
    assert torch.equal(words, w2)
    # a single invalid byte deep inside 10^9 bases is found exactly
    seq[987_654_321] = ord("N")
    torch.cuda.synchronize()
    ctx.encode_dev(seq, n, words)
    import bitnuc_amd as bn
    with pytest.raises(bn.NucleotideError) as ei:
        ctx.sync()
    assert (ei.value.byte, ei.value.index) == (ord("N"), 987_654_321)


def test_beyond_4gib_indexing(ctx, oracle):
    """2^32 + 12345 bases: byte offsets and group indices past 32 bits (encode, decode, scan, error index)."""
    import torch
    import bitnuc_amd as bn
    dev = torch.device("cuda:0")
    n = (1 << 32) + 12345
    nw = (n + 31) // 32
    seq = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.nucgen_dev(seq, n, 0xB17C0DE)
    words = torch.empty(nw, dtype=torch.int64, device=dev)
    back = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.encode_dev(seq, n, words)
    ctx.decode_dev(words, nw, n, back)
    ctx.sync()
    assert torch.equal(seq, back)
    for off in (0, (1 << 32) - 64 * 1000, n - 64 * 1000 - (n % 32)):
        m = min(64 * 1000 + 32, n - off)
        h = seq[off:off + m].cpu().numpy()
        assert np.array_equal(h, oracle.nucgen(m, 0xB17C0DE, first=off))
        assert np.array_equal(words[off // 32: off // 32 + (m + 31) // 32].cpu().numpy().view(np.uint64), oracle.encode(h))
    del back
    k = 31
    q = oracle.as_2bit(seq[n - 500:n - 500 + k].cpu().numpy())
    dist = torch.empty(n - k + 1, dtype=torch.uint8, device=dev)
    ctx.kmer_hdist_scan_dev(seq, n, k, q, dist)
    ctx.sync()
    assert int(dist[n - 500]) == 0
    off = (1 << 32) - 3000
    h = seq[off:off + 6000 + k - 1].cpu().numpy()
    assert np.array_equal(dist[off:off + 6000].cpu().numpy(), oracle.kmer_hdist_scan(h, k, q))
    tail = seq[n - 4000:].cpu().numpy()
    assert np.array_equal(dist[n - 4000:].cpu().numpy(), oracle.kmer_hdist_scan(tail, k, q))
    del dist
    pos = (1 << 32) + 77
    seq[pos] = ord("N")
    torch.cuda.synchronize()
    ctx.encode_dev(seq, n, words)
    with pytest.raises(bn.NucleotideError) as ei:
        ctx.sync()
    assert (ei.value.byte, ei.value.index) == (ord("N"), pos)


def test_beyond_4gib_batches(ctx, oracle):
    """Byte offsets past 2^32 in the batch kernels: dense and strided k-mer batches, fixed-length reads, ragged batches."""
    import torch
    dev = torch.device("cuda:0")
    n = (1 << 32) + 150 * 1000
    seq = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.nucgen_dev(seq, n, 0xB17C0DE)
    ctx.sync()
    # k-mer batches: dense (stride == k) and strided, k-mers that start beyond 2^32
    for k, stride in ((31, 31), (31, 40)):
        count = (n - k) // stride + 1
        out = torch.empty(count, dtype=torch.int64, device=dev)
        ctx.as_2bit_batch_dev(seq, k, stride, count, out)
        ctx.sync()
        for j in (0, (1 << 32) // stride - 2, (1 << 32) // stride + 3, count - 1):
            h = seq[j * stride: j * stride + k].cpu().numpy()
            assert int(out[j].item()) & (2**64 - 1) == oracle.as_2bit(h), (k, stride, j)
        del out
    # fixed-length reads == ragged batch of the same reads; decode of both is the input
    L = 150
    count = n // L
    wpr = (L + 31) // 32
    words = torch.empty(count * wpr, dtype=torch.int64, device=dev)
    ctx.encode_fixed_dev(seq, L, L, count, words)
    off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L
    wo = torch.empty(count + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    total = ctx.batch_word_offsets_dev(off, count, wo)
    assert total == count * wpr
    w2 = torch.empty(total, dtype=torch.int64, device=dev)
    ctx.encode_batch_dev(seq, off, wo, count, total, w2)
    ctx.sync()
    assert torch.equal(words, w2)
    r = (1 << 32) // L  # the read that straddles byte 2^32
    for rr in (r - 1, r, r + 1, count - 1):
        h = seq[rr * L:(rr + 1) * L].cpu().numpy()
        assert np.array_equal(words[rr * wpr:(rr + 1) * wpr].cpu().numpy().view(np.uint64), oracle.encode(h)), rr
    del w2
    back = torch.empty(count * L, dtype=torch.uint8, device=dev)
    ctx.decode_fixed_dev(words, L, L, count, back)
    ctx.sync()
    assert torch.equal(back, seq[: count * L])
    back.zero_()
    torch.cuda.synchronize()
    ctx.decode_batch_dev(words, wo, off, count, total, back)
    ctx.sync()
    assert torch.equal(back, seq[: count * L])
    # the layout plan on the same batch: tile bases past 2^32, same words, same bases
    import bitnuc_amd as bn
    plan = bn.BatchPlan(ctx, off, count)
    assert plan.total_words == total
    w3 = torch.zeros(total, dtype=torch.int64, device=dev)
    back.zero_()
    torch.cuda.synchronize()
    plan.encode_dev(seq, w3)
    plan.decode_dev(w3, back)
    ctx.sync()
    assert torch.equal(w3, words) and torch.equal(back, seq[: count * L])
    plan.close()


def test_batch_fuzz_vs_oracle_loop(ctx, sweep_ctx, oracle):
    """Random ragged batches (length mixes incl. empties and sub-word sequences, a batch that starts anywhere in its buffer,
    lower case) through the plan and the table-driven kernels, against the oracle's per-sequence loop; then one invalid byte
    at a random position: (byte, index) of the first one in buffer order."""
    import bitnuc_amd as bn
    rng = np.random.default_rng(20260)
    mixes = [lambda m: rng.integers(0, 4, size=m), lambda m: rng.integers(1, 70, size=m), lambda m: rng.integers(100, 260, size=m),
             lambda m: np.where(rng.random(m) < 0.3, 0, rng.integers(1, 40, size=m)), lambda m: rng.integers(1, 5000, size=m),
             lambda m: np.where(rng.random(m) < 0.9, 32, rng.integers(0, 3, size=m)), lambda m: np.full(m, 64) + rng.integers(0, 2, size=m)]
    for trial in range(28):
        lengths = [int(x) for x in mixes[trial % len(mixes)](int(rng.integers(1, 1500)))]
        seq, off = _ragged(lengths)
        pre = int(rng.integers(0, 40))
        buf = np.concatenate([np.full(pre, ord("N"), np.uint8), seq, np.full(int(rng.integers(0, 20)), ord("N"), np.uint8)])
        off2 = off + np.uint64(pre)
        ew, ewo = _oracle_batch(oracle, seq, off)
        product = ctx
        for use_plan in (1, 0):
            ctx = product if use_plan else sweep_ctx  # host calls through the table-driven form: a switch of the evidence build
            prev = ctx.set_variant("batch_host_plan", use_plan)
            try:
                w, wo = ctx.encode_batch(buf, off2)
                assert np.array_equal(wo, ewo) and np.array_equal(w, ew), (trial, use_plan)
                back = ctx.decode_batch(w, wo, off2)
                assert bytes(back[pre:pre + len(seq)]) == bytes(seq).upper(), (trial, use_plan)
                if len(seq):
                    bad = buf.copy()
                    pos = pre + int(rng.integers(0, len(seq)))
                    later = pre + int(rng.integers(pos - pre, len(seq)))
                    bad[later] = ord("X")
                    bad[pos] = 0xFF if trial & 1 else ord("n")
                    with pytest.raises(bn.NucleotideError) as ei:
                        ctx.encode_batch(bad, off2)
                    assert (ei.value.kind, ei.value.byte, ei.value.index) == ("InvalidBase", int(bad[pos]), pos), (trial, use_plan)
            finally:
                ctx.set_variant("batch_host_plan", prev)


def test_batch_tables_at_odd_word_addresses(ctx, oracle):
    """The offsets / word-offsets tables only have to be 8-byte aligned: device tables that start one entry into an allocation (the scan
    kernels load them 16 bytes at a time), counts around the scan's 8-per-thread and 2048-per-workgroup granularity."""
    import torch
    import bitnuc_amd as bn
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(77)
    for count in (1, 7, 8, 9, 2047, 2048, 2049, 5000, 16385):
        lengths = rng.integers(0, 90, size=count)
        off = np.zeros(count + 1, dtype=np.int64)
        off[1:] = np.cumsum(lengths)
        seq = rand_seq(int(off[-1]) + 1)
        d_seq = torch.from_numpy(seq).to(dev)
        hold = torch.zeros(count + 2, dtype=torch.int64, device=dev)
        hold[1:] = torch.from_numpy(off).to(dev)
        d_off = hold[1:]                       # 8 bytes into the allocation
        assert d_off.data_ptr() % 16 == 8
        wo_hold = torch.zeros(count + 2, dtype=torch.int64, device=dev)
        d_wo = wo_hold[1:]
        torch.cuda.synchronize()
        total = ctx.batch_word_offsets_dev(d_off, count, d_wo)
        exp_wo = np.zeros(count + 1, dtype=np.int64)
        exp_wo[1:] = np.cumsum((lengths + 31) // 32)
        assert total == int(exp_wo[-1]) and np.array_equal(d_wo.cpu().numpy(), exp_wo), count
        ew, _ = _oracle_batch(oracle, seq[: int(off[-1])], off.astype(np.uint64))
        plan = bn.BatchPlan(ctx, d_off, count)
        assert plan.total_words == total
        words = torch.zeros(max(total, 1), dtype=torch.int64, device=dev)
        back = torch.zeros(int(off[-1]) + 1, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        plan.encode_dev(d_seq, words)
        plan.decode_dev(words, back)
        ctx.sync()
        assert np.array_equal(words[:total].cpu().numpy().view(np.uint64), ew), count
        assert bytes(back[: int(off[-1])].cpu().numpy()) == bytes(seq[: int(off[-1])]).upper(), count
        plan.close()


def test_hbm_scale_round_trip(ctx, oracle):
    """One sequence sized for the 288 GB of HBM: up to 10^11 bases (100 GB ASCII + 25 GB packed + 100 GB decoded),
    scaled down to what the box has free.  Round-trip identity, spot blocks against the oracle, error index."""
    import torch
    import bitnuc_amd as bn
    dev = torch.device("cuda:0")
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    n = min(10**11, int(free * 0.40)) // 32 * 32 + 17
    while True:  # a fragmented or shared device: halve until the three buffers fit (the test is about scale, not a fixed size)
        nw = (n + 31) // 32
        try:
            seq = torch.empty(n, dtype=torch.uint8, device=dev)
            words = torch.empty(nw, dtype=torch.int64, device=dev)
            back = torch.empty(n, dtype=torch.uint8, device=dev)
            break
        except torch.OutOfMemoryError:
            seq = words = back = None
            torch.cuda.empty_cache()
            n = n // 2 // 32 * 32 + 17
            assert n > 10**8, "not even 10^8 bases fit"
    torch.cuda.synchronize()
    ctx.nucgen_dev(seq, n, 0xB17C0DE)
    ctx.encode_dev(seq, n, words)
    ctx.decode_dev(words, nw, n, back)
    ctx.sync()
    step = 1 << 31  # torch.equal materialises a temporary as large as its operands: compare 2 GiB at a time
    assert all(torch.equal(seq[i:i + step], back[i:i + step]) for i in range(0, n, step))
    del back
    for off in (0, n // 3 // 32 * 32, n - 32 * 1000 - 17):
        m = n - off if off + 32 * 1000 + 17 == n else 32 * 1000  # whole words, except at the very end (17-base tail)
        h = seq[off:off + m].cpu().numpy()
        assert np.array_equal(h, oracle.nucgen(m, 0xB17C0DE, first=off))
        assert np.array_equal(words[off // 32: off // 32 + (m + 31) // 32].cpu().numpy().view(np.uint64), oracle.encode(h))
    pos = n - 12345
    seq[pos] = ord("n")
    torch.cuda.synchronize()
    ctx.encode_dev(seq, n, words)
    with pytest.raises(bn.NucleotideError) as ei:
        ctx.sync()
    assert (ei.value.byte, ei.value.index) == (ord("n"), pos)
    del seq, words
    torch.cuda.empty_cache()


def test_context_churn_releases_device_memory(oracle):
    """Create / use / destroy many contexts: every device allocation of a context (slots, accumulators, scratch that
    grew to 64 MiB) is released with it."""
    import torch
    import bitnuc_amd as bn
    s = rand_seq(1 << 22)
    exp = oracle.encode(s)
    for _ in range(12):  # the runtime keeps a bounded pool of hardware queues for non-blocking streams (~150 MiB, reached
        bn.Context(0).close()  # after < 10 streams): fill it first, it is not the contexts' memory
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for i in range(40):
        c = bn.Context(0)
        assert np.array_equal(c.encode_array(s), exp)
        assert c.base_counts(exp, s.size) == oracle.base_counts(exp, s.size)
        big = rand_seq(1 << 26) if i == 0 else None  # grow the staging scratch once per loop head
        if big is not None:
            c.encode_array(big)
        c.close()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (64 << 20), f"{(free0 - free1) >> 20} MiB not returned"


def test_two_contexts_interleaved(oracle):
    import threading
    import bitnuc_amd as bn
    errs = []

    def work(seed):
        try:
            c = bn.Context(0)
            rng = np.random.default_rng(seed)
            for _ in range(20):
                n = int(rng.integers(1, 200000))
                s = ALPHA8[rng.integers(0, 8, size=n)]
                w = c.encode_array(s)
                assert np.array_equal(w, oracle.encode(s))
                assert bytes(c.decode_array(w, n)) == bytes(s).upper()
            c.close()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs


# BASELINE configs 3 and 5 at full size: tests/test_gpu_round4.py compares EVERY output element (closed form / whole-input oracle run);
# the sampled comparison that stood here is gone with it.


def test_cpp_host_layer(ctx):
    """include/bitnuc.hpp (the compiled mirror of the reference's Rust API) against the
    reference's own unit-test cases, via tests/cpp/test_bitnuc_hpp.cpp."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "test_bitnuc_hpp")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "cpp", "test_bitnuc_hpp.cpp"),
                    "-L", os.path.join(root, "bitnuc_amd"), "-lbitnuc_hip",
                    "-Wl,-rpath," + os.path.join(root, "bitnuc_amd"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL OK" in out.stdout


# ---- analysis on packed words -----------------------------------------------
def test_analysis_golden_and_oracle(ctx, golden, oracle):
    import bitnuc_amd as bn
    for v in golden["gc_content"]:
        s = v["seq"].encode()
        assert ctx.gc_content(ctx.encode_alloc(s), len(s)) == v["gc"], v["src"]
    for v in golden["base_counts"]:
        s = v["seq"].encode()
        assert ctx.base_counts(ctx.encode_alloc(s), len(s)) == v["counts"], v["src"]
    e = golden["empty_sequence_analysis"]
    assert ctx.gc_content([], 0) == e["gc"] and ctx.base_counts([], 0) == e["counts"]
    for n in [1, 31, 32, 33, 63, 64, 65, 1000, 100003, 3000001]:
        s = rand_seq(n)
        w = oracle.encode(s)
        assert ctx.base_counts(w, n) == oracle.base_counts(w, n), n
        assert ctx.gc_content(w, n) == oracle.gc_content(w, n), n
        # bits above n_bases in the last word are ignored
        w2 = w.copy()
        if n % 32:
            w2[-1] |= np.uint64(0xFFFFFFFFFFFFFFFF) << np.uint64(2 * (n % 32))
        assert ctx.base_counts(w2, n) == oracle.base_counts(w, n), n
    with pytest.raises(bn.NucleotideError) as ei:
        ctx.base_counts(np.zeros(1, np.uint64), 33)
    assert ei.value.kind == "InvalidLength"


def test_hdist_pairs_and_query(ctx, oracle):
    import bitnuc_amd as bn
    for count in [1, 3, 4, 5, 1023, 1024, 100003]:
        a = RNG.integers(0, 1 << 63, size=count, dtype=np.uint64) * np.uint64(2) + RNG.integers(0, 2, size=count, dtype=np.uint64)
        b = RNG.integers(0, 1 << 63, size=count, dtype=np.uint64) * np.uint64(2) + RNG.integers(0, 2, size=count, dtype=np.uint64)
        for length in (0, 1, 15, 16, 17, 31, 32):
            assert np.array_equal(ctx.hdist_pairs(a, b, length), oracle.hdist_pairs(a, b, length)), (count, length)
            q = int(b[0])
            assert np.array_equal(ctx.hdist_query(q, a, length), oracle.hdist_pairs(a, np.full(count, q, np.uint64), length)), (count, length)
    with pytest.raises(bn.NucleotideError) as ei:
        ctx.hdist_pairs([0], [0], 33)
    assert ei.value.kind == "InvalidLength" and ei.value.len == 33


def test_invalid_base_reported_after_input_is_overwritten(ctx):
    """The latched error carries the byte itself: the input may be reused before the sync."""
    import torch
    import bitnuc_amd as bn
    dev = torch.device("cuda:0")
    n = 1_000_003
    seq = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.nucgen_dev(seq, n, 7)
    ctx.sync()
    seq[777_777] = ord("N")
    seq[900_000] = ord("X")
    torch.cuda.synchronize()
    words = torch.empty((n + 31) // 32, dtype=torch.int64, device=dev)
    dist = torch.empty(n - 30, dtype=torch.uint8, device=dev)
    launches = [lambda t: ctx.encode_dev(t, n, words), lambda t: ctx.kmer_hdist_scan_dev(t, n, 31, 0, dist),
                lambda t: ctx.as_2bit_batch_dev(t, 31, 31, n // 31, words), lambda t: ctx.as_2bit_batch_dev(t, 31, 37, n // 37, words),
                lambda t: ctx.encode_fixed_dev(t, 150, 151, n // 151, words)]
    for launch in launches:
        work = seq.clone()
        torch.cuda.synchronize()
        launch(work)
        ctx.nucgen_dev(work, n, 9)  # stream-ordered overwrite of the input, before anything is reported
        with pytest.raises(bn.NucleotideError) as ei:
            ctx.sync()
        assert (ei.value.kind, ei.value.byte, ei.value.index) == ("InvalidBase", ord("N"), 777_777)


def test_split_packed_golden(ctx, oracle, golden):
    import bitnuc_amd as bn
    for canonical in (False, True):
        for v in golden["split_packed"]:
            s = v["seq"].encode()
            ebuf, lbuf, rbuf = [], [123], [456]  # cleared by the call, split.rs:31-32
            ctx.encode(s, ebuf)
            ctx.split_packed(ebuf, len(s), v["idx"], lbuf, rbuf, canonical=canonical)
            if not canonical:
                assert (len(lbuf), len(rbuf)) == (v["n_left"], v["n_right"]), v["src"]
            left, right = bytearray(), bytearray()
            ctx.decode(lbuf, len(v["left"]), left)
            ctx.decode(rbuf, len(v["right"]), right)
            assert (bytes(left), bytes(right)) == (v["left"].encode(), v["right"].encode()), v["src"]
        e = golden["split_packed_err"]
        lbuf, rbuf = [1], [2]
        with pytest.raises(bn.NucleotideError) as ei:
            ctx.split_packed(oracle.encode(e["seq"].encode()), len(e["seq"]), e["idx"], lbuf, rbuf, canonical=canonical)
        assert ei.value == bn.NucleotideError("IndexOutOfBounds", index=e["index"], length=e["length"])
        assert (lbuf, rbuf) == ([1], [2])  # validated before the buffers are cleared, split.rs:23-32


def test_split_packed_vs_oracle(ctx, oracle):
    import bitnuc_amd as bn
    for n in [1, 31, 32, 33, 64, 65, 200, 1000, 4099, 100003]:
        s = rand_seq(n, ALPHA)
        w = oracle.encode(s)
        idxs = sorted({0, 1, n // 3, n // 2, (n // 2) & ~31, n - 1, n} | {int(x) for x in RNG.integers(0, n + 1, size=6)})
        for idx in idxs:
            lbuf, rbuf = [], []
            ctx.split_packed(w, n, idx, lbuf, rbuf)  # the reference word for word
            lo, ro = oracle.split_packed(w, n, idx)
            assert np.array_equal(np.array(lbuf, dtype=np.uint64), lo), (n, idx)
            assert np.array_equal(np.array(rbuf, dtype=np.uint64), ro), (n, idx)
            assert ctx.split_packed_sizes(w.size, n, idx) == (lo.size, ro.size)
            ctx.split_packed(w, n, idx, lbuf, rbuf, canonical=True)  # == encode of the two halves
            assert np.array_equal(np.array(lbuf, dtype=np.uint64), oracle.encode(s[:idx]) if idx else np.zeros(0, np.uint64)), (n, idx)
            assert np.array_equal(np.array(rbuf, dtype=np.uint64), oracle.encode(s[idx:]) if idx < n else np.zeros(0, np.uint64)), (n, idx)
    # device buffers at 8-byte (not 16-byte) alignment take the one-word-per-lane path
    import torch
    dev = torch.device("cuda:0")
    n = 70001
    s = rand_seq(n, ALPHA)
    w = oracle.encode(s)
    buf = torch.zeros(w.size + 1, dtype=torch.int64, device=dev)
    buf[1:] = torch.from_numpy(w.view(np.int64)).to(dev)
    for idx in (37, 32 * 1001 + 9, 32 * 1002 + 9, 64 * 500):
        for canonical in (False, True):
            nl, nr = ctx.split_packed_sizes(w.size, n, idx, canonical=canonical)
            for off in (0, 1):  # all operands misaligned, then only the source
                lt, rt = torch.zeros(nl + 1, dtype=torch.int64, device=dev), torch.zeros(nr + 1, dtype=torch.int64, device=dev)
                torch.cuda.synchronize()  # the fills run on torch's stream, the split on the context's
                ctx.split_packed_dev(buf[1:], w.size, n, idx, lt[1 - off:], rt[1 - off:], canonical=canonical)
                ctx.sync()
                got_l = lt[1 - off: 1 - off + nl].cpu().numpy().view(np.uint64)
                got_r = rt[1 - off: 1 - off + nr].cpu().numpy().view(np.uint64)
                exp_l, exp_r = (oracle.encode(s[:idx]), oracle.encode(s[idx:])) if canonical else oracle.split_packed(w, n, idx)
                assert np.array_equal(got_l, exp_l) and np.array_equal(got_r, exp_r), (idx, canonical, off)
    # junk above the last base is cleared in canonical mode and carried along as written
    w = np.array([2**64 - 1, 2**64 - 1], dtype=np.uint64)
    lbuf, rbuf = [], []
    ctx.split_packed(w, 40, 7, lbuf, rbuf, canonical=True)
    assert lbuf == [(1 << 14) - 1] and rbuf == [2**64 - 1, 3]
    ctx.split_packed(w, 40, 7, lbuf, rbuf)
    lo, ro = oracle.split_packed(w, 40, 7)
    assert lbuf == [int(x) for x in lo] and rbuf == [int(x) for x in ro]
    # short buffers: defined as InvalidLength(slen) (the reference panics / data-dependent length)
    for canonical in (False, True):
        with pytest.raises(bn.NucleotideError) as ei:
            ctx.split_packed(w, 200, 100, lbuf, rbuf, canonical=canonical)
        assert ei.value == bn.NucleotideError("InvalidLength", len=200)
    ctx.split_packed([], 10, 5, lbuf, rbuf)  # empty ebuf, split.rs:47-49
    assert (lbuf, rbuf) == ([], [])


def test_split_packed_full_scale(ctx, oracle):
    """10^9+17 bases on the device: canonical split at an odd base == encode of the two halves."""
    import torch
    dev = torch.device("cuda:0")
    n = 10**9 + 17
    idx = 333_333_341
    nw = (n + 31) // 32
    seq = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.nucgen_dev(seq, n, 0xB17C0DE)
    words = torch.empty(nw, dtype=torch.int64, device=dev)
    ctx.encode_dev(seq, n, words)
    nl, nr = ctx.split_packed_sizes(nw, n, idx, canonical=True)
    assert (nl, nr) == ((idx + 31) // 32, (n - idx + 31) // 32)
    left, right = torch.empty(nl, dtype=torch.int64, device=dev), torch.empty(nr, dtype=torch.int64, device=dev)
    ctx.split_packed_dev(words, nw, n, idx, left, right, canonical=True)
    el, er = torch.empty(nl, dtype=torch.int64, device=dev), torch.empty(nr, dtype=torch.int64, device=dev)
    ctx.encode_dev(seq, idx, el)
    ctx.encode_dev(seq[idx:], n - idx, er)  # unaligned device pointer
    ctx.sync()
    assert torch.equal(left, el) and torch.equal(right, er)
    # as written: sizes and a sampled window against the oracle's restatement
    nl, nr = ctx.split_packed_sizes(nw, n, idx)
    assert (nl, nr) == (idx // 32 + 1, nw - idx // 32)
    left, right = torch.empty(nl, dtype=torch.int64, device=dev), torch.empty(nr, dtype=torch.int64, device=dev)
    ctx.split_packed_dev(words, nw, n, idx, left, right)
    ctx.sync()
    c, s = idx // 32, (idx % 32) * 2
    wh = words[c - 1: c + 1001].cpu().numpy().view(np.uint64)
    exp = [(int(wh[j + 1]) >> s) | ((int(wh[j]) << (64 - s)) & (2**64 - 1) if j else 0) for j in range(1000)]
    assert [int(x) for x in right[:1000].cpu().numpy().view(np.uint64)] == exp
    assert torch.equal(left[: c], words[: c]) and int(left[c]) == int(wh[1]) & ((1 << s) - 1)


def test_hdist_u32_accumulator_wraps_like_the_reference(ctx, oracle):
    """hamming/multi.rs:130 sums into a u32: 2^32 + 5 mismatching bases give 5 (release-build wrap-around).
    Base counts of the same buffers stay exact in u64."""
    import torch
    dev = torch.device("cuda:0")
    n = (1 << 32) + 5
    nw = (n + 31) // 32
    a = torch.zeros(nw, dtype=torch.int64, device=dev)       # all A
    t = torch.full((nw,), -1, dtype=torch.int64, device=dev)  # all T (pad bits set too: they must be ignored)
    res = torch.zeros(1, dtype=torch.int32, device=dev)
    counts = torch.zeros(4, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.hdist_dev(a, nw, t, nw, n, res)
    ctx.base_counts_dev(t, nw, n, counts)
    ctx.sync()
    assert int(res.item()) & 0xFFFFFFFF == 5
    assert counts.tolist() == [0, 0, 0, n]
    # the oracle's accumulator wraps the same way (checked on a slice that crosses 2^32 only in the sum:
    # 2^27 + 1 words of 32 mismatches each would take the scalar loop a second; the closed form suffices)
    assert (32 * (1 << 27) + 5) % (1 << 32) == 5 and oracle.hdist(np.zeros(2, np.uint64), np.full(2, 2**64 - 1, np.uint64), 37) == 37


def test_analysis_full_scale(ctx, oracle):
    import torch
    dev = torch.device("cuda:0")
    n = 10**9 + 17
    nw = (n + 31) // 32
    seq = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.nucgen_dev(seq, n, 0xB17C0DE)
    words = torch.empty(nw, dtype=torch.int64, device=dev)
    ctx.encode_dev(seq, n, words)
    counts = torch.zeros(4, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()  # the fill runs on torch's stream, the kernels on the context's
    ctx.base_counts_dev(words, nw, n, counts)
    ctx.sync()
    expect = [int((seq == ord(c)).sum()) for c in "ACGT"]  # independent: counted on the ASCII side by torch
    assert counts.tolist() == expect and sum(expect) == n
    # one query against 3.1e7 packed 32-mers
    dist = torch.empty(nw, dtype=torch.uint8, device=dev)
    q = int(words[12345].item()) & 0xFFFFFFFFFFFFFFFF
    ctx.hdist_query_dev(q, words, nw - 1, 32, dist)
    ctx.sync()
    assert int(dist[12345]) == 0
    h = words[:5000].cpu().numpy().view(np.uint64)
    assert np.array_equal(dist[:5000].cpu().numpy(), oracle.hdist_pairs(h, np.full(5000, q, np.uint64), 32))


# ---- PackedSequence: the reference's own tests, src/sequence.rs:266-338 --------
def test_packed_sequence_golden(ctx, golden, oracle):
    import bitnuc_amd as bn
    g = golden["packed_sequence"]
    seq = bn.PackedSequence.new(g["new"]["seq"].encode(), ctx)
    assert seq.len() == g["new"]["len"] and len(seq) == g["new"]["len"] and seq.to_vec() == g["new"]["to_vec"].encode()
    seq = bn.PackedSequence(g["get"]["seq"].encode(), ctx)
    assert [chr(seq.get(i)) for i in range(4)] == g["get"]["bases"]
    with pytest.raises(bn.NucleotideError) as ei:
        seq.get(g["get_oob"]["index"])
    assert ei.value == bn.NucleotideError("IndexOutOfBounds", index=4, length=4)
    for v in g["slices"]:
        assert bn.PackedSequence(v["seq"].encode(), ctx).slice(v["start"], v["end"]) == v["out"].encode(), v["src"]
    v = g["invalid_slice"]
    with pytest.raises(bn.NucleotideError) as ei:
        bn.PackedSequence(v["seq"].encode(), ctx).slice(v["start"], v["end"])
    assert ei.value == bn.NucleotideError("InvalidRange", start=3, end=2, length=4)
    a, b = (bn.PackedSequence(s.encode(), ctx) for s in g["equality"]["same"])
    c = bn.PackedSequence(g["equality"]["different"][1].encode(), ctx)
    assert a == b and a != c and b in {a} and c not in {a}
    with pytest.raises(bn.NucleotideError):
        bn.PackedSequence(g["invalid"]["seq"].encode(), ctx)
    e = bn.PackedSequence(b"", ctx)
    assert e.is_empty() and e.len() == 0 and e.to_vec() == b"" and e.gc_content() == 0.0 and e.base_counts() == [0, 0, 0, 0]
    for v in golden["gc_content"]:
        assert bn.PackedSequence(v["seq"].encode(), ctx).gc_content() == v["gc"]
    for v in golden["base_counts"]:
        assert bn.PackedSequence(v["seq"].encode(), ctx).base_counts() == v["counts"]
    # random slices against the ASCII original
    s = rand_seq(5000)
    up = bytes(s).upper()
    p = bn.PackedSequence(s, ctx)
    assert p.to_vec() == up
    for _ in range(50):
        i, j = sorted(int(x) for x in RNG.integers(0, 5001, size=2))
        assert p.slice(i, j) == up[i:j]
    assert all(p.get(i) == up[i] for i in (0, 1, 31, 32, 33, 4999))


def test_crate_integration_cases(ctx, golden):
    """src/lib.rs:222-265 (the crate's own end-to-end tests) and src/sequence.rs:328-338 through the Python mirror."""
    import bitnuc_amd as bn
    g = golden["crate_integration"]
    v = g["creation_and_analysis"]
    seq = bn.PackedSequence.new(v["seq"].encode(), ctx)
    assert seq.len() == v["len"] and seq.is_empty() == v["is_empty"] and seq.to_vec() == v["to_vec"].encode()
    assert seq.gc_content() == v["gc"] and seq.base_counts() == v["counts"]
    v = g["mutations"]
    seq = bn.PackedSequence.new(v["seq"].encode(), ctx)
    assert seq.slice(v["slice"][0], v["slice"][1]) == v["slice"][2].encode()
    assert all(seq.get(i) == ord(b) for i, b in v["get"])
    v = g["error_handling"]
    with pytest.raises(bn.NucleotideError):
        bn.PackedSequence.new(v["invalid"].encode(), ctx)
    seq = bn.PackedSequence.new(v["seq"].encode(), ctx)
    with pytest.raises(bn.NucleotideError) as ei:
        seq.get(v["get_oob"])
    assert ei.value.kind == "IndexOutOfBounds"
    with pytest.raises(bn.NucleotideError) as ei:
        seq.slice(*v["slice_oob"])
    assert ei.value.kind == "InvalidRange"
    v = g["hashability"]
    held = {bn.PackedSequence.new(v["in_set"][0].encode(), ctx)}
    assert bn.PackedSequence.new(v["in_set"][1].encode(), ctx) in held
    assert bn.PackedSequence.new(v["not_in_set"].encode(), ctx) not in held


def test_hip_graph_capture_of_a_step(oracle):
    """The _dev entry points allocate nothing and never synchronise, so an encode+decode step
    can be captured into a hipGraph and replayed on new data (launch-bound pipelines)."""
    import torch
    import bitnuc_amd as bn
    dev = torch.device("cuda:0")
    n = 1_000_003
    nw = (n + 31) // 32
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        c = bn.Context(0, stream=s.cuda_stream)
        seq = torch.empty(n, dtype=torch.uint8, device=dev)
        words = torch.empty(nw, dtype=torch.int64, device=dev)
        back = torch.empty(n, dtype=torch.uint8, device=dev)
        c.nucgen_dev(seq, n, 1)
        c.encode_dev(seq, n, words)
        c.decode_dev(words, nw, n, back)
        c.sync()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            c.encode_dev(seq, n, words)
            c.decode_dev(words, nw, n, back)
        for seed in (2, 3, 4):
            c.nucgen_dev(seq, n, seed)
            back.zero_()
            g.replay()
            s.synchronize()
            assert torch.equal(seq, back)
            h = seq[:64000].cpu().numpy()
            assert np.array_equal(words[:2000].cpu().numpy().view(np.uint64), oracle.encode(h))
        # a replay on invalid data latches the error in the captured launch's slot
        seq[777] = ord("N")
        g.replay()
        with pytest.raises(bn.NucleotideError) as ei:
            c.sync()
        assert (ei.value.byte, ei.value.index) == (ord("N"), 777)
        c.close()


def test_rccl_comm_single_rank(oracle):
    """C-ABI RCCL path on the one GPU we have: a 1-rank communicator (all-gather == copy) through
    init_rank, and the single-process init_all / encode_sharded_allgather_all form with n = 1."""
    import ctypes as C
    import torch
    import bitnuc_amd as bn
    from bitnuc_amd import _lib as L
    dev = torch.device("cuda:0")
    c = bn.Context(0)
    comm = bn.Comm(c, 1, 0, bn.Comm.unique_id())
    assert (comm.nranks, comm.rank) == (1, 0)
    n = 32 * 40001
    seq = torch.empty(n, dtype=torch.uint8, device=dev)
    c.nucgen_dev(seq, n, 5)
    allw = torch.zeros(n // 32, dtype=torch.int64, device=dev)
    c.sync()
    torch.cuda.synchronize()
    comm.encode_sharded_allgather_dev(seq, n, allw)
    c.sync()
    assert np.array_equal(allw.cpu().numpy().view(np.uint64), oracle.encode(seq.cpu().numpy()))
    out = torch.zeros(n // 32, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    comm.allgather_words_dev(allw, n // 32, out)
    c.sync()
    assert torch.equal(out, allw)
    with pytest.raises(bn.NucleotideError) as ei:
        comm.encode_sharded_allgather_dev(seq, n - 1, allw)  # shards must be whole words
    assert ei.value.kind == "InvalidLength"
    comm.close()
    c.close()
    # single-process form
    lib = L.load()
    ctxs, comms = (C.c_void_p * 1)(), (C.c_void_p * 1)()
    err = L.BitnucErr()
    assert lib.bitnuc_comm_init_all(1, ctxs, comms, C.byref(err)) == 0, err.backend_code
    seqs = (C.c_void_p * 1)(seq.data_ptr())
    alls = (C.c_void_p * 1)(out.data_ptr())
    out.zero_()
    torch.cuda.synchronize()
    assert lib.bitnuc_encode_sharded_allgather_all(1, ctxs, comms, seqs, n, alls, C.byref(err)) == 0, err.backend_code
    assert torch.equal(out, allw)
    assert lib.bitnuc_comm_single_process(comms[0]) == 1
    for chunks in (1, 5):  # the chunked form, driven for all (one) ranks by this thread
        out.zero_()
        torch.cuda.synchronize()
        assert lib.bitnuc_encode_sharded_allgather_overlapped_all(1, ctxs, comms, seqs, n, chunks, alls, C.byref(err)) == 0, err.backend_code
        assert torch.equal(out, allw), chunks
    seq[n - 7] = ord("N")
    torch.cuda.synchronize()
    assert lib.bitnuc_encode_sharded_allgather_overlapped_all(1, ctxs, comms, seqs, n, 4, alls, C.byref(err)) == L.INVALID_BASE
    assert (err.byte, err.index, err.value) == (ord("N"), n - 7, 0)
    assert lib.bitnuc_encode_sharded_allgather_overlapped_all(1, ctxs, comms, seqs, n - 1, 4, alls, C.byref(err)) == L.INVALID_LENGTH
    # a one-rank single-process communicator has no peer to wait for: the per-rank form is allowed on it
    assert lib.bitnuc_encode_sharded_allgather_overlapped_dev(ctxs[0], comms[0], seq.data_ptr(), 32, 1, out.data_ptr(), C.byref(err)) == 0
    assert lib.bitnuc_ctx_sync(ctxs[0], C.byref(err)) == 0
    lib.bitnuc_comm_destroy(comms[0])
    lib.bitnuc_ctx_destroy(ctxs[0])


def test_encode_many_and_fuzz(ctx, oracle):
    """Seeded fuzz over lengths, alphabets, strides and batch shapes (every result vs the oracle)."""
    rng = np.random.default_rng(20251004)
    seqs = [bytes(rand_seq(int(n))) for n in rng.integers(1, 300, size=200)]
    got = ctx.encode_many(seqs)
    for s, w in zip(seqs, got):
        assert np.array_equal(w, oracle.encode(s))
    for _ in range(150):
        n = int(rng.choice([rng.integers(1, 70), rng.integers(70, 5000), rng.integers(5000, 400000)]))
        s = ALPHA8[rng.integers(0, 8, size=n)]
        w = ctx.encode_array(s)
        assert np.array_equal(w, oracle.encode(s)), n
        m = int(rng.integers(1, n + 1))
        assert np.array_equal(ctx.decode_array(w, m), oracle.decode(w, m)), (n, m)
        k = int(rng.integers(1, 33))
        if n >= k:
            stride = int(rng.integers(1, 70))
            count = (n - k) // stride + 1
            assert np.array_equal(ctx.as_2bit_batch(s, k, stride, count), oracle.as_2bit_batch(s, k, stride, count)), (n, k, stride)
            q = int(rng.integers(0, 1 << 63)) * 2 + int(rng.integers(0, 2))
            assert np.array_equal(ctx.kmer_hdist_scan(s, k, q), oracle.kmer_hdist_scan(s, k, q)), (n, k)
        # ragged split of the same bytes into random pieces, incl. empty ones
        cuts = np.sort(rng.integers(0, n + 1, size=int(rng.integers(0, 40))))
        off = np.concatenate([[0], cuts, [n]]).astype(np.uint64)
        bw, bwo = ctx.encode_batch(s, off)
        ew, ewo = _oracle_batch(oracle, s, off)
        assert np.array_equal(bw, ew) and np.array_equal(bwo, ewo), n
        assert bytes(ctx.decode_batch(bw, bwo, off)) == bytes(s).upper()


def test_ballot_formulation_variant(sweep_ctx, oracle):
    """north_star's lane-per-base + wavefront-ballot encode (variant 100): same bits, same errors."""
    import bitnuc_amd as bn
    ctx = sweep_ctx
    prev = ctx.set_variant("encode", 100)
    try:
        for n in [1, 31, 32, 33, 63, 64, 65, 127, 128, 1000, 4097, 1000003]:
            s = rand_seq(n)
            assert np.array_equal(ctx.encode_array(s), oracle.encode(s)), n
        s = rand_seq(100000).copy()
        s[54321] = ord("N")
        s[70000] = ord("X")
        with pytest.raises(bn.NucleotideError) as ei:
            ctx.encode_array(s)
        assert (ei.value.byte, ei.value.index) == (ord("N"), 54321)
    finally:
        ctx.set_variant("encode", prev)


# ---- fixed-length reads --------------------------------------------------------------------------
@pytest.mark.parametrize("read_len,stride", [(150, 150), (150, 151), (100, 100), (31, 31), (32, 32), (33, 40), (1, 1), (1, 3),
                                             (250, 250), (250, 251), (151, 300), (64, 200), (1000, 1000), (5000, 5003)])
def test_fixed_reads_vs_oracle_loop(ctx, oracle, read_len, stride):
    import bitnuc_amd as bn
    for count in [1, 2, 13, 64, 1000]:
        nbytes = (count - 1) * stride + read_len
        buf = np.full(nbytes, ord("\n"), dtype=np.uint8)  # separators are never examined
        for r in range(count):
            buf[r * stride: r * stride + read_len] = rand_seq(read_len)
        exp = np.stack([oracle.encode(buf[r * stride: r * stride + read_len]) for r in range(count)])
        got = ctx.encode_fixed(buf, read_len, stride, count)
        assert got.shape == exp.shape and np.array_equal(got, exp), (read_len, stride, count)
        back = np.full(nbytes, ord("#"), dtype=np.uint8)
        ctx.decode_fixed(got, read_len, stride, out=back)
        for r in (0, count // 2, count - 1):
            assert bytes(back[r * stride: r * stride + read_len]) == bytes(buf[r * stride: r * stride + read_len]).upper()
        if stride > read_len and count > 1:
            assert bytes(back[read_len:stride]) == b"#" * (stride - read_len)  # gap bytes untouched
    # an invalid base inside a read is reported with its offset in the caller's buffer
    count = 500
    buf = np.full((count - 1) * stride + read_len, ord("\n"), dtype=np.uint8)
    for r in range(count):
        buf[r * stride: r * stride + read_len] = rand_seq(read_len)
    pos = 321 * stride + read_len - 1
    buf[pos] = ord("N")
    with pytest.raises(bn.NucleotideError) as ei:
        ctx.encode_fixed(buf, read_len, stride, count)
    assert (ei.value.byte, ei.value.index) == (ord("N"), pos)


@pytest.mark.parametrize("body", [2, 1], ids=["shared-tile-body", "strip-64bit-positions"])
def test_decode_fixed_contiguous_every_alignment(ctx, sweep_ctx, oracle, body):
    """Back-to-back reads: whole output compared, at several output alignments, with junk in the pad bits of each
    read's last word, and with guard bytes around the run."""
    import torch
    dev = torch.device("cuda:0")
    if body != 2:
        ctx = sweep_ctx  # the earlier kernel exists in the evidence build only
    prev = ctx.set_variant("fixed_dec_strip", body)
    try:
        for read_len, count in [(16, 700), (17, 333), (31, 500), (32, 129), (33, 257), (47, 100), (64, 65), (100, 1001), (150, 777),
                                (250, 300), (2048, 9), (2049, 9), (5000, 7), (100003, 3), (20, 1), (150, 1)]:
            wpr = (read_len + 31) // 32
            seq = rand_seq(read_len * count, ALPHA)
            words = np.stack([oracle.encode(seq[r * read_len:(r + 1) * read_len]) for r in range(count)]).reshape(-1).copy()
            rem = read_len % 32
            if rem:  # bits above a read's last base are ignored by decode (from_2bit's expected_size semantics)
                words[wpr - 1::wpr] |= np.uint64((0xDEADBEEFCAFEF00D << (2 * rem)) & (2**64 - 1))
            d_words = torch.from_numpy(words.view(np.int64)).to(dev)
            for a in (0, 1, 7, 8, 15):
                buf = torch.full((read_len * count + 64,), ord("#"), dtype=torch.uint8, device=dev)
                torch.cuda.synchronize()  # the fill runs on torch's stream, the decode on the context's
                ctx.decode_fixed_dev(d_words, read_len, read_len, count, buf[16 + a:])
                ctx.sync()
                got = buf.cpu().numpy()
                assert bytes(got[:16 + a]) == b"#" * (16 + a), (read_len, count, a)
                assert bytes(got[16 + a + read_len * count:]) == b"#" * (48 - a), (read_len, count, a)
                assert np.array_equal(got[16 + a: 16 + a + read_len * count], seq), (read_len, count, a)
    finally:
        ctx.set_variant("fixed_dec_strip", prev)


def test_fixed_reads_full_scale(ctx, oracle):
    import torch
    dev = torch.device("cuda:0")
    L, count = 150, 6_666_666
    n = L * count
    seq = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.nucgen_dev(seq, n, 0xB17C0DE)
    words = torch.empty(count * 5, dtype=torch.int64, device=dev)
    back = torch.zeros(n, dtype=torch.uint8, device=dev)
    ctx.sync()
    ctx.encode_fixed_dev(seq, L, L, count, words)
    ctx.decode_fixed_dev(words, L, L, count, back)
    ctx.sync()
    assert torch.equal(seq, back)
    # the ragged-batch path must give the same words
    off = torch.arange(0, count + 1, dtype=torch.int64, device=dev) * L
    wo = torch.empty(count + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    total = ctx.batch_word_offsets_dev(off, count, wo)
    w2 = torch.empty(total, dtype=torch.int64, device=dev)
    ctx.encode_batch_dev(seq, off, wo, count, total, w2)
    ctx.sync()
    assert torch.equal(words, w2)
    h = seq[:L * 1000].cpu().numpy()
    exp = np.concatenate([oracle.encode(h[i * L:(i + 1) * L]) for i in range(1000)])
    assert np.array_equal(words[:5000].cpu().numpy().view(np.uint64), exp)
