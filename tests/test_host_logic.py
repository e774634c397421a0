"""CPU tests of the host-side mirror that need no GPU: error vocabulary and Display strings
(src/error.rs:3-45), status mapping of the ctypes layer."""
import pytest


def test_nucleotide_error_display_matches_reference():
    from bitnuc_amd import NucleotideError as E
    # format strings of src/error.rs:22-44
    assert str(E("InvalidBase", byte=78)) == "Invalid nucleotide base: 78"
    assert str(E("SequenceTooLong", len=33)) == "Sequence length 33 exceeds maximum"
    assert str(E("InvalidLength", len=40)) == "Invalid length: 40"
    assert str(E("IndexOutOfBounds", index=4, length=4)) == "Index 4 out of bounds for sequence of length 4"
    assert str(E("InvalidRange", start=3, end=2, length=4)) == "Invalid range 3..2 for sequence of length 4"
    assert str(E("Unsupported")) == "Unsupported architecture"


def test_nucleotide_error_equality_is_variant_plus_payload():
    from bitnuc_amd import NucleotideError as E
    assert E("InvalidBase", byte=78) == E("InvalidBase", byte=78, index=123)  # index is an extra, not part of the Rust variant
    assert E("InvalidBase", byte=78) != E("InvalidBase", byte=79)
    assert E("SequenceTooLong", len=33) != E("InvalidLength", len=33)
    assert E("IndexOutOfBounds", index=4, length=4) != E("IndexOutOfBounds", index=5, length=4)


def test_status_mapping():
    from bitnuc_amd import _lib as L
    from bitnuc_amd import api
    for status, kind in [(L.INVALID_BASE, "InvalidBase"), (L.SEQUENCE_TOO_LONG, "SequenceTooLong"),
                         (L.INVALID_LENGTH, "InvalidLength"), (L.UNSUPPORTED, "Unsupported")]:
        err = L.BitnucErr()
        err.status, err.byte, err.value, err.index = status, 78, 33, 7
        with pytest.raises(api.NucleotideError) as ei:
            api._raise(err)
        assert ei.value.kind == kind
    err = L.BitnucErr()
    err.status, err.backend_code = L.BACKEND_ERROR, 100
    with pytest.raises(api.BackendError):
        api._raise(err)
