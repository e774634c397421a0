"""Property tests of PackedSequence (the reference's owned type, src/sequence.rs:5-262, and its analysis traits,
src/utils/analysis.rs:3-39) on inputs small enough for the library's host code: no GPU needed (except for the analysis traits, which are kernels only).  The model is the definition:
base i of the sequence is code (data[i / 32] >> 2 (i % 32)) & 3, rendered upper case."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import bitnuc_amd as bn
from bitnuc_amd import api
from bitnuc_amd.sequence import PackedSequence as _PackedSequence


def PackedSequence(seq):  # a NULL context: below the host cutoff every call is the library's host code (SURVEY 8b size dispatch)
    from bitnuc_amd import build
    build.ensure_built()
    return _PackedSequence(seq, ctx=api.context_free())

bases = st.text(alphabet="ACGTacgt", min_size=0, max_size=700).map(lambda s: s.encode())


@settings(max_examples=150, deadline=None)
@given(bases, st.data())
def test_get_slice_to_vec_follow_the_definition(seq, data):
    p = PackedSequence(seq)
    up = seq.upper()
    assert len(p) == p.len() == len(seq) and p.is_empty() == (len(seq) == 0)
    assert p.data.size == (len(seq) + 31) // 32  # sequence.rs:42-46: no words for an empty sequence
    assert p.to_vec() == up
    if seq:
        i = data.draw(st.integers(0, len(seq) - 1))
        assert p.get(i) == up[i]  # sequence.rs:116-135
        a = data.draw(st.integers(0, len(seq)))
        b = data.draw(st.integers(a, len(seq)))
        assert p.slice(a, b) == up[a:b]  # sequence.rs:198-212
        # the unused high bits of the last word are zero (packing/naive.rs:17: only `len` shifts are ORed in)
        tail = len(seq) % 32
        if tail:
            assert int(p.data[-1]) >> (2 * tail) == 0
    # out of range: the reference's error values (sequence.rs:117-119, :199-205)
    with pytest.raises(bn.NucleotideError) as e:
        p.get(len(seq))
    assert e.value == bn.NucleotideError("IndexOutOfBounds", index=len(seq), length=len(seq))
    with pytest.raises(bn.NucleotideError) as e:
        p.slice(0, len(seq) + 1)
    assert e.value == bn.NucleotideError("InvalidRange", start=0, end=len(seq) + 1, length=len(seq))
    if len(seq) >= 2:
        with pytest.raises(bn.NucleotideError) as e:
            p.slice(2, 1)
        assert e.value.kind == "InvalidRange"


@settings(max_examples=100, deadline=None)
@given(bases, bases)
def test_equality_and_hash_are_those_of_words_and_length(a, b):
    pa, pb = PackedSequence(a), PackedSequence(b)
    assert (pa == pb) == (a.upper() == b.upper())  # derive(PartialEq) on (data, length): case is not stored
    if pa == pb:
        assert hash(pa) == hash(pb)
    assert pa == PackedSequence(a.lower())


@pytest.mark.gpu
@settings(max_examples=100, deadline=None)
@given(seq=bases)
def test_analysis_traits_count_what_decoding_would_count(ctx, seq):
    """The counts run on the packed words on the GPU (there is no host code for them): every input here is a launch."""
    p = _PackedSequence(seq, ctx=ctx)
    up = seq.upper()
    assert p.base_counts() == [up.count(c) for c in (b"A", b"C", b"G", b"T")]  # analysis.rs:23-39
    want = 0.0 if not seq else (up.count(b"G") + up.count(b"C")) / len(seq) * 100.0  # analysis.rs:7-16
    assert p.gc_content() == want


@settings(max_examples=60, deadline=None)
@given(st.binary(min_size=1, max_size=200))
def test_new_reports_the_first_invalid_base(raw):
    bad = [i for i, c in enumerate(raw) if c not in b"ACGTacgt"]
    if not bad:
        assert PackedSequence(raw).to_vec() == raw.upper()
        return
    with pytest.raises(bn.NucleotideError) as e:
        PackedSequence(raw)
    assert e.value == bn.NucleotideError("InvalidBase", byte=raw[bad[0]])  # packing/avx.rs:86-91: the first one of the whole sequence
