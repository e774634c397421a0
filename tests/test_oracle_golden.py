"""CPU tests: pin the oracle (oracle/bitnuc_oracle.c and the AVX2 restatement) to
every known-answer vector the reference's own tests hold (tests/golden/golden.json),
and cross-check the two restatements against each other."""
import numpy as np
import pytest

RNG = np.random.default_rng(0xB17C0DE)
ALPHA = np.frombuffer(b"ACGT", dtype=np.uint8)
ALPHA8 = np.frombuffer(b"ACGTacgt", dtype=np.uint8)


def rand_seq(n, alpha=ALPHA):
    return alpha[RNG.integers(0, len(alpha), size=n)]


def test_as_2bit_vectors(oracle, golden):
    for v in golden["as_2bit"]:
        assert oracle.as_2bit(v["seq"].encode()) == v["packed"], v["src"]
    ci = golden["as_2bit_case_insensitive"]
    assert oracle.as_2bit(ci["lower"].encode()) == oracle.as_2bit(ci["upper"].encode())
    assert oracle.as_2bit(b"") == 0


def test_as_2bit_errors(oracle, golden):
    for v in golden["as_2bit_err"]:
        with pytest.raises(oracle.OracleError) as ei:
            oracle.as_2bit(v["seq"].encode())
        assert ei.value.kind == v["status"], v["src"]
        if "byte" in v:
            assert ei.value.byte == v["byte"]
        if "value" in v:
            assert ei.value.value == v["value"]
    # length is checked before any base (naive.rs:5-7): 33 invalid bytes -> TooLong
    with pytest.raises(oracle.OracleError) as ei:
        oracle.as_2bit(b"N" * 33)
    assert ei.value.kind == "SequenceTooLong"


def test_from_2bit_vectors(oracle, golden):
    for v in golden["from_2bit"]:
        assert oracle.from_2bit(v["packed"], v["n"]) == v["seq"].encode(), v["src"]
    for v in golden["from_2bit_err"]:
        with pytest.raises(oracle.OracleError) as ei:
            oracle.from_2bit(v["packed"], v["n"])
        assert ei.value.kind == v["status"] and ei.value.value == v["value"]
    for v in golden["from_2bit_append"]:
        p = oracle.as_2bit(v["seq"].encode())
        buf = bytearray()
        for _ in range(v["calls"]):
            buf += oracle.from_2bit(p, v["n"])
        assert bytes(buf) == v["expected"].encode(), v["src"]
    v = golden["from_2bit_simd20"]
    assert oracle.from_2bit(oracle.as_2bit(v["seq"].encode()), v["n"]) == v["seq"].encode()


def test_roundtrips(oracle, golden):
    for s in golden["roundtrip_strings"]["cases"]:
        b = s.encode()
        assert oracle.from_2bit(oracle.as_2bit(b), len(b)) == b
    rp = golden["roundtrip_prefixes"]
    for n in range(rp["lens"][0], rp["lens"][1] + 1):
        b = rp["seq"].encode()[:n]
        assert oracle.from_2bit(oracle.as_2bit(b), n) == b
    lo, hi = golden["roundtrip_lengths"]["lens"]
    for n in range(lo, hi + 1):  # src/utils/mod.rs:113-133 (BASELINE config 1 at n = 1000)
        s = rand_seq(n)
        for avx2 in (False, True):
            w = oracle.encode(s, avx2=avx2)
            assert w.size == (n + 31) // 32
            assert np.array_equal(oracle.decode(w, n, avx2=avx2), s)


def test_hdist_vectors(oracle, golden):
    for v in golden["hdist_scalar"]:
        assert oracle.hdist_scalar(v["u"], v["v"], v["len"]) == v["d"], v["src"]
    for a, b, d in golden["hdist_scalar_strings"]["cases"]:
        assert oracle.hdist_scalar(oracle.as_2bit(a.encode()), oracle.as_2bit(b.encode()), len(a)) == d
    for v in golden["hdist_scalar_err"]:
        with pytest.raises(oracle.OracleError) as ei:
            oracle.hdist_scalar(v["u"], v["v"], v["len"])
        assert ei.value.kind == v["status"] and ei.value.value == v["value"]
    for v in golden["hdist_err"]:
        with pytest.raises(oracle.OracleError) as ei:
            oracle.hdist(np.zeros(v["na"], np.uint64), np.zeros(v["nb"], np.uint64), v["n_bases"])
        assert ei.value.kind == v["status"] and ei.value.value == v["value"]
    for v in golden["hdist"]:
        a, b = oracle.encode(v["seq1"].encode()), oracle.encode(v["seq2"].encode())
        assert oracle.hdist(a, b, len(v["seq1"])) == v["d"], v["src"]
    lo, hi = golden["hdist_A_vs_T"]["lens"]
    for n in range(lo, hi + 1):
        assert oracle.hdist(oracle.encode(b"A" * n), oracle.encode(b"T" * n), n) == n
    for v in golden["hdist_cyclic"]:
        s1 = np.array([ALPHA[i % v["mod1"]] for i in range(v["l"])], dtype=np.uint8)
        s2 = np.array([ALPHA[i % v["mod2"]] for i in range(v["l"])], dtype=np.uint8)
        expect = int((s1 != s2).sum())
        assert oracle.hdist(oracle.encode(s1), oracle.encode(s2), v["l"]) == expect, v["src"]
        if v["l"] <= 32:
            assert oracle.hdist_scalar(oracle.as_2bit(s1), oracle.as_2bit(s2), v["l"]) == expect


def test_kmer_count_doc_example(oracle, golden):
    v = golden["kmer_count"]
    seq = v["seq"].encode()
    words = oracle.as_2bit_batch(seq, v["k"], 1, len(seq) - v["k"] + 1)
    assert int((words == oracle.as_2bit(v["kmer"].encode())).sum()) == v["count"]


def test_encode_error_semantics(oracle):
    # first invalid byte of the WHOLE sequence; ebuf keeps the words of the chunks before it
    s = rand_seq(200).copy()
    s[77] = ord("N")
    s[150] = ord("X")
    for avx2 in (False, True):
        with pytest.raises(oracle.OracleError) as ei:
            oracle.encode(s, avx2=avx2)
        e = ei.value
        assert (e.kind, e.byte, e.index) == ("InvalidBase", ord("N"), 77)
        assert e.words.size == 77 // 32
        assert np.array_equal(e.words, oracle.encode(s[:64]))
    # empty input: the reference panics (avx.rs:138)
    with pytest.raises(oracle.OracleError) as ei:
        oracle.encode(b"")
    assert ei.value.kind == "Panic"


def test_decode_short_buffer(oracle):
    with pytest.raises(oracle.OracleError) as ei:
        oracle.decode(np.zeros(1, np.uint64), 33)
    assert ei.value.kind == "InvalidLength" and ei.value.value == 33
    assert oracle.decode(np.zeros(0, np.uint64), 0).size == 0


def test_avx2_restatement_equals_scalar(oracle):
    for n in [1, 15, 16, 17, 31, 32, 33, 47, 48, 63, 64, 65, 1000, 4099, 100003]:
        s = rand_seq(n, ALPHA8)
        a, b = oracle.encode(s), oracle.encode(s, avx2=True)
        assert np.array_equal(a, b), n
        assert np.array_equal(oracle.decode(a, n), oracle.decode(a, n, avx2=True))
    # invalid byte at every offset class, incl. the scalar tail of a 16..31-byte chunk
    for n in [20, 40, 100]:
        for pos in range(n):
            for bad in (ord("N"), 0, 0xFF, ord("@"), ord("B"), ord("U")):
                s = rand_seq(n, ALPHA8).copy()
                s[pos] = bad
                errs = []
                for avx2 in (False, True):
                    with pytest.raises(oracle.OracleError) as ei:
                        oracle.encode(s, avx2=avx2)
                    errs.append((ei.value.kind, ei.value.byte, ei.value.index, ei.value.words.tobytes()))
                assert errs[0] == errs[1] == ("InvalidBase", bad, pos, errs[0][3])


def test_scan_equals_composition(oracle):
    s = rand_seq(500, ALPHA8)
    for k in (1, 4, 15, 16, 31, 32):
        q = oracle.as_2bit(rand_seq(k))
        d = oracle.kmer_hdist_scan(s, k, q)
        assert d.size == 500 - k + 1
        up = np.frombuffer(bytes(s).upper(), dtype=np.uint8)
        qs = np.frombuffer(oracle.from_2bit(q, k), dtype=np.uint8)
        for i in (0, 1, 17, 100, 500 - k):
            assert d[i] == int((up[i:i + k] != qs).sum())
    with pytest.raises(oracle.OracleError) as ei:
        oracle.kmer_hdist_scan(s, 33, 0)
    assert ei.value.kind == "SequenceTooLong"
    assert oracle.kmer_hdist_scan(s[:10], 31, 0).size == 0


def test_nucgen_properties(oracle):
    a = oracle.nucgen(1000, 0xB17C0DE)
    assert set(np.unique(a)) <= set(ALPHA)
    # stream is position-addressable: any slice regenerates identically
    assert np.array_equal(oracle.nucgen(300, 0xB17C0DE, first=123), a[123:423])
    cyc = oracle.nucgen(64, 0, first=5, flags=1)
    assert bytes(cyc[:8]) == b"CGTACGTA"
    # not degenerate
    counts = np.bincount(oracle.nucgen(1 << 16, 7), minlength=128)[ALPHA]
    assert counts.min() > (1 << 16) / 4 * 0.95


def test_analysis_vectors(oracle, golden):
    for v in golden["gc_content"]:
        s = v["seq"].encode()
        assert oracle.gc_content(oracle.encode(s), len(s)) == v["gc"], v["src"]
    for v in golden["base_counts"]:
        s = v["seq"].encode()
        assert oracle.base_counts(oracle.encode(s), len(s)) == v["counts"], v["src"]
    e = golden["empty_sequence_analysis"]
    assert oracle.gc_content(np.zeros(0, np.uint64), 0) == e["gc"]
    assert oracle.base_counts(np.zeros(0, np.uint64), 0) == e["counts"]
    s = rand_seq(1000)
    w = oracle.encode(s)
    assert oracle.base_counts(w, 1000) == [int((s == ord(c)).sum()) for c in "ACGT"]
    a, b = RNG.integers(0, 1 << 63, size=100, dtype=np.uint64), RNG.integers(0, 1 << 63, size=100, dtype=np.uint64)
    d = oracle.hdist_pairs(a, b, 31)
    assert all(int(d[i]) == oracle.hdist_scalar(int(a[i]), int(b[i]), 31) for i in range(100))


def test_split_packed_vectors(oracle, golden):
    from oracle_py import OracleError
    for v in golden["split_packed"]:
        s = v["seq"].encode()
        lo, ro = oracle.split_packed(oracle.encode(s), len(s), v["idx"])
        assert (lo.size, ro.size) == (v["n_left"], v["n_right"]), v["src"]
        assert oracle.decode(lo, len(v["left"])).tobytes() == v["left"].encode(), v["src"]
        assert oracle.decode(ro, len(v["right"])).tobytes() == v["right"].encode(), v["src"]
    e = golden["split_packed_err"]
    with pytest.raises(OracleError) as ei:
        oracle.split_packed(oracle.encode(e["seq"].encode()), len(e["seq"]), e["idx"])
    assert (ei.value.kind, ei.value.index, ei.value.value) == (e["status"], e["index"], e["length"])
    # as written: a split inside the last word, or on a word boundary, gives the shifted sequence ...
    s = rand_seq(200)
    w = oracle.encode(s)
    for idx in (32, 64, 160, 192, 195, 199):
        lo, ro = oracle.split_packed(w, 200, idx)
        assert oracle.decode(lo, idx).tobytes() == s[:idx].tobytes()
        assert oracle.decode(ro, 200 - idx).tobytes() == s[idx:].tobytes()
    # ... and a longer one with a shift carries the previous word's low bits (split.rs:84-94)
    lo, ro = oracle.split_packed(w, 200, 5)
    assert lo.size == 1 and ro.size == 7
    assert int(ro[0]) == int(w[0]) >> 10 and int(ro[1]) == ((int(w[1]) >> 10) | ((int(w[0]) << 54) & (2**64 - 1)))
    # a buffer that does not reach the split word panics (split.rs:78)
    with pytest.raises(OracleError) as ei:
        oracle.split_packed(w[:2], 200, 100)
    assert ei.value.kind == "Panic"
