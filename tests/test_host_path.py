"""CPU tests (no GPU) of the library's HOST path -- SURVEY 8(b): `as_2bit` / `from_2bit` / `hdist_scalar`
and bulk host-pointer calls below the host cutoff are the library's own SWAR code (bitnuc_amd/csrc/host_word.h),
reachable with a NULL context.  Checked against the reference's golden vectors and, exhaustively, against the oracle:
all 256 byte values x positions 0..31, every length 0..=32, every length 1..=1000 round trip, first-invalid-byte
ordering, the Vec-truncation rule, and the refusal of large inputs without a device."""
import numpy as np
import pytest

import bitnuc_amd
from bitnuc_amd import api


@pytest.fixture(scope="module")
def host():
    from bitnuc_amd import build
    build.ensure_built()
    return api.context_free()


RNG = np.random.default_rng(20261004)


def rand_seq(n, lower=0.25):
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[RNG.integers(0, 4, size=n)]
    return np.where(RNG.random(n) < lower, s | 0x20, s).astype(np.uint8)


def test_golden_as_2bit_from_2bit(host, golden):
    for v in golden["as_2bit"]:
        assert host.as_2bit(v["seq"].encode()) == v["packed"], v
    for v in golden["from_2bit"]:
        assert host.from_2bit_alloc(v["packed"], v["n"]) == v["seq"].encode(), v
    for v in golden["as_2bit_err"]:
        with pytest.raises(bitnuc_amd.NucleotideError) as ei:
            host.as_2bit(v["seq"].encode())
        assert ei.value.kind == v["status"], v
    for v in golden["from_2bit_err"]:
        with pytest.raises(bitnuc_amd.NucleotideError) as ei:
            host.from_2bit_alloc(v["packed"], v["n"])
        assert ei.value.kind == v["status"] and ei.value.len == v["value"], v
    v = golden["from_2bit_append"][0]
    buf = bytearray()
    for _ in range(v["calls"]):
        host.from_2bit(host.as_2bit(v["seq"].encode()), v["n"], buf)
    assert bytes(buf) == v["expected"].encode()
    v = golden["roundtrip_prefixes"]
    for n in range(v["lens"][0], v["lens"][1] + 1):
        assert host.from_2bit_alloc(host.as_2bit(v["seq"][:n].encode()), n) == v["seq"][:n].encode()
    for sq in golden["roundtrip_strings"]["cases"]:
        assert host.from_2bit_alloc(host.as_2bit(sq.encode()), len(sq)) == sq.encode()
    with pytest.raises(bitnuc_amd.NucleotideError) as ei:
        host.as_2bit(b"ACGN")  # packing/mod.rs:186-187
    assert ei.value.kind == "InvalidBase" and ei.value.byte == ord("N") and ei.value.index == 3
    with pytest.raises(bitnuc_amd.NucleotideError) as ei:
        host.as_2bit(b"A" * 33)  # packing/mod.rs:192-196
    assert ei.value.kind == "SequenceTooLong" and ei.value.len == 33
    with pytest.raises(bitnuc_amd.NucleotideError) as ei:
        host.from_2bit_alloc(0, 33)  # unpacking/mod.rs:95-98
    assert ei.value.kind == "InvalidLength" and ei.value.len == 33
    assert host.as_2bit(b"") == 0
    assert host.as_2bit(b"acgt") == host.as_2bit(b"ACGT") == 0xE4


def test_module_level_single_word_functions_need_no_device(golden):
    assert bitnuc_amd.as_2bit(b"ACGT") == 0b11100100  # README.md:25-26
    assert bitnuc_amd.from_2bit_alloc(0xE4, 4) == b"ACGT"
    buf = bytearray(b"xx")
    bitnuc_amd.from_2bit(0xE4, 2, buf)  # appends (unpacking/avx.rs:185-194)
    assert bytes(buf) == b"xxAC"
    assert bitnuc_amd.hdist_scalar(bitnuc_amd.as_2bit(b"ACGT"), bitnuc_amd.as_2bit(b"ACGA"), 4) == 1


def test_all_256_byte_values_at_every_position(host, oracle):
    base = b"ACGTACGTACGTACGTACGTACGTACGTACGT"
    for pos in range(32):
        for b in range(256):
            s = bytearray(base)
            s[pos] = b
            try:
                exp = oracle.as_2bit(bytes(s))
            except oracle.OracleError as e:
                with pytest.raises(bitnuc_amd.NucleotideError) as ei:
                    host.as_2bit(bytes(s))
                assert (ei.value.kind, ei.value.byte, ei.value.index) == (e.kind, b, pos)
                continue
            assert host.as_2bit(bytes(s)) == exp, (pos, b)


def test_every_length_and_first_invalid_order(host, oracle):
    for n in range(0, 33):
        s = rand_seq(n)
        w = host.as_2bit(s)
        assert w == oracle.as_2bit(s)
        assert host.from_2bit_alloc(w, n) == oracle.from_2bit(w, n)
        # bits above 2n are ignored by from_2bit (unpacking/naive.rs:12-21)
        assert host.from_2bit_alloc(w | (0xFFFFFFFFFFFFFFFF << (2 * n)) & 0xFFFFFFFFFFFFFFFF if n < 32 else w, n) == oracle.from_2bit(w, n)
    s = bytearray(rand_seq(32, lower=0))
    s[9], s[20] = ord("N"), ord("X")
    with pytest.raises(bitnuc_amd.NucleotideError) as ei:
        host.as_2bit(bytes(s))
    assert (ei.value.byte, ei.value.index) == (ord("N"), 9)


def test_hdist_scalar_golden_and_random(host, oracle, golden):
    for v in golden["hdist_scalar"]:
        assert host.hdist_scalar(v["u"], v["v"], v["len"]) == v["d"], v
    for a, b, d in golden["hdist_scalar_strings"]["cases"]:
        assert host.hdist_scalar(host.as_2bit(a.encode()), host.as_2bit(b.encode()), len(a)) == d
    with pytest.raises(bitnuc_amd.NucleotideError) as ei:
        host.hdist_scalar(0, 0, 33)  # hamming/scalar.rs:55-60
    assert ei.value.kind == "InvalidLength"
    for _ in range(2000):
        u, v = int(RNG.integers(0, 1 << 63)) * 2 + int(RNG.integers(0, 2)), int(RNG.integers(0, 1 << 63)) * 2 + 1
        n = int(RNG.integers(0, 33))
        assert host.hdist_scalar(u, v, n) == oracle.hdist_scalar(u, v, n)


def test_bulk_below_cutoff_roundtrip_all_lengths(host, oracle):
    # src/utils/mod.rs:113-133: every length 1..=1000 (BASELINE configs[0] is the 1000-base case)
    for n in list(range(1, 1001)) + [1023, 1024, 1025, 4097, 65535]:
        s = rand_seq(n)
        w = host.encode_array(s)
        assert np.array_equal(w, oracle.encode(s)), n
        d = host.decode_array(w, n)
        assert np.array_equal(d, oracle.decode(w, n)), n
        assert bytes(d) == bytes(s).upper()


def test_bulk_invalid_base_semantics(host, oracle):
    for n, bad_at in [(1000, 0), (1000, 15), (1000, 16), (1000, 31), (1000, 32), (1000, 999), (100, 64), (33, 32)]:
        for byte in (ord("N"), 0x00, 0xFF, ord("n"), ord("@")):
            s = rand_seq(n).copy()
            s[bad_at] = byte
            if bad_at + 40 < n:
                s[bad_at + 40] = ord("X")  # a later invalid byte must not win
            with pytest.raises(oracle.OracleError) as eo:
                oracle.encode(s)
            with pytest.raises(bitnuc_amd.NucleotideError) as ei:
                host.encode_array(s)
            assert (ei.value.kind, ei.value.byte, ei.value.index) == ("InvalidBase", byte, bad_at)
            assert eo.value.byte == byte
            # the Vec holds the words of the chunks before the failing one (packing/avx.rs:142-143)
            assert np.array_equal(ei.value.words, eo.value.words)


def test_decode_short_buffer_and_hdist(host, oracle, golden):
    with pytest.raises(bitnuc_amd.NucleotideError) as ei:
        host.decode_array(np.zeros(1, dtype=np.uint64), 64)  # unpacking/mod.rs:40-45
    assert ei.value.kind == "InvalidLength" and ei.value.len == 64
    for n in [1, 31, 32, 33, 64, 255, 256, 1000]:
        a, b = rand_seq(n), rand_seq(n)
        wa, wb = host.encode_array(a), host.encode_array(b)
        assert host.hdist(wa, wb, n) == oracle.hdist(wa, wb, n) == int((np.char.upper(a.view("S1")) != np.char.upper(b.view("S1"))).sum())
    with pytest.raises(bitnuc_amd.NucleotideError):
        host.hdist(np.zeros(1, dtype=np.uint64), np.zeros(1, dtype=np.uint64), 64)  # hamming/multi.rs:167-172


def test_large_inputs_need_a_device(host):
    # at or above the cutoff a NULL context is refused: bulk work is the kernels' and there is no CPU fallback for it
    s = np.full(1 << 20, ord("A"), dtype=np.uint8)  # the encode cutoff (the measured host / GPU crossover)
    with pytest.raises(bitnuc_amd.NucleotideError) as ei:
        host.encode_array(s)
    assert ei.value.kind == "Unsupported"
    w = host.encode_array(s[:-1])
    assert w.size == (s.size - 1 + 31) // 32
    with pytest.raises(bitnuc_amd.NucleotideError) as ei:  # decode's cutoff is lower: 512 Ki bases
        host.decode_array(w, 1 << 19)
    assert ei.value.kind == "Unsupported"
    assert bytes(host.decode_array(w, (1 << 19) - 1)) == b"A" * ((1 << 19) - 1)


def test_packed_sequence_get_is_shift_and_mask():
    # sequence.rs:116-135 -- no library call, so it works on hand-made words without any device
    from bitnuc_amd.sequence import PackedSequence
    p = PackedSequence.__new__(PackedSequence)
    p.data, p.length, p._ctx = np.array([0xE4, 0x1B], dtype=np.uint64), 36, None
    assert bytes(p.get(i) for i in range(4)) == b"ACGT"
    assert bytes(p.get(32 + i) for i in range(4)) == b"TGCA"
    with pytest.raises(bitnuc_amd.NucleotideError):
        p.get(36)


def test_selftime_small_runs(host):
    from bitnuc_amd import _lib
    lib = _lib.load()
    for op, n in [(0, 31), (1, 32), (2, 1000), (3, 1000), (4, 32)]:
        ns = lib.bitnuc_selftime_small(op, n, 2000)
        assert 0 < ns < 1e6
    assert lib.bitnuc_selftime_small(0, 33, 10) < 0
