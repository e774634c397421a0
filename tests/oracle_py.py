"""ctypes wrapper of oracle/libbitnuc_oracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference (oracle/bitnuc_oracle.c,
oracle/bitnuc_avx2.c).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; nothing under bitnuc_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libbitnuc_oracle.so")

STATUS = {0: "Ok", 1: "InvalidBase", 2: "SequenceTooLong", 3: "InvalidLength", 4: "IndexOutOfBounds", 99: "Panic"}


class OrcErr(C.Structure):
    _fields_ = [("status", C.c_int32), ("byte", C.c_uint8), ("value", C.c_uint64), ("index", C.c_uint64)]


class OracleError(Exception):
    def __init__(self, err):
        self.kind = STATUS.get(err.status, str(err.status))
        self.byte, self.value, self.index = int(err.byte), int(err.value), int(err.index)
        super().__init__(f"{self.kind} byte={self.byte} value={self.value} index={self.index}")


_lib = None


def build():
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("bitnuc_oracle.c", "bitnuc_avx2.c", "bitnuc_oracle.h", "Makefile")]
    if not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.run(["make", "-C", ORACLE_DIR, "libbitnuc_oracle.so"], check=True, capture_output=True)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        P, SZ, U64 = C.c_void_p, C.c_size_t, C.c_uint64
        E = C.POINTER(OrcErr)
        L.orc_as_2bit.argtypes = [P, SZ, C.POINTER(U64), E]
        L.orc_from_2bit.argtypes = [U64, SZ, P, E]
        L.orc_encode.argtypes = [P, SZ, P, C.POINTER(SZ), E]
        L.orc_decode.argtypes = [P, SZ, SZ, P, E]
        L.orc_hdist_scalar.argtypes = [U64, U64, SZ, C.POINTER(C.c_uint32), E]
        L.orc_hdist.argtypes = [P, SZ, P, SZ, SZ, C.POINTER(C.c_uint32), E]
        L.orc_as_2bit_batch.argtypes = [P, SZ, SZ, SZ, P, E]
        L.orc_kmer_hdist_scan.argtypes = [P, SZ, SZ, U64, P, E]
        L.orc_base_counts.argtypes = [P, SZ, SZ, P, E]
        L.orc_gc_content.argtypes = [P, SZ, SZ]
        L.orc_gc_content.restype = C.c_double
        L.orc_hdist_pairs.argtypes = [P, P, SZ, SZ, P, E]
        L.orc_split_packed.argtypes = [P, SZ, SZ, SZ, P, C.POINTER(SZ), P, C.POINTER(SZ), E]
        L.orc_nucgen.argtypes = [P, SZ, U64, U64, C.c_int]
        L.orc_nucgen.restype = None
        L.orc_avx2_encode.argtypes = [P, SZ, C.POINTER(P), C.POINTER(SZ), E]
        L.orc_avx2_decode.argtypes = [P, SZ, SZ, C.POINTER(P), C.POINTER(SZ), E]
        L.orc_free.argtypes = [P]
        L.orc_free.restype = None
        L.orc_avx2_time_roundtrip.argtypes = [P, SZ, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _lib = L
    return _lib


def _u8(x):
    a = np.frombuffer(x, dtype=np.uint8) if not isinstance(x, np.ndarray) else x
    return np.ascontiguousarray(a, dtype=np.uint8)


def _p(a):
    return C.c_void_p(a.ctypes.data if a.size else 0)


def as_2bit(seq):
    s = _u8(seq)
    out, err = C.c_uint64(0), OrcErr()
    if lib().orc_as_2bit(_p(s), s.size, C.byref(out), C.byref(err)):
        raise OracleError(err)
    return out.value


def from_2bit(packed, n):
    out, err = np.zeros(32, dtype=np.uint8), OrcErr()
    if lib().orc_from_2bit(C.c_uint64(packed), n, _p(out), C.byref(err)):
        raise OracleError(err)
    return out[:n].tobytes()


def encode(seq, avx2=False):
    """-> ndarray[uint64]; raises OracleError with .words = words pushed before the failure."""
    s = _u8(seq)
    err = OrcErr()
    if avx2:
        ptr, nw = C.c_void_p(), C.c_size_t(0)
        st = lib().orc_avx2_encode(_p(s), s.size, C.byref(ptr), C.byref(nw), C.byref(err))
        words = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(nw.value,)).copy() if nw.value else np.zeros(0, np.uint64)
        lib().orc_free(ptr)
    else:
        out = np.zeros((s.size + 31) // 32, dtype=np.uint64)
        nw = C.c_size_t(0)
        st = lib().orc_encode(_p(s), s.size, _p(out), C.byref(nw), C.byref(err))
        words = out[: nw.value]
    if st:
        e = OracleError(err)
        e.words = words
        raise e
    return words


def decode(ebuf, n_bases, avx2=False):
    e = np.ascontiguousarray(ebuf, dtype=np.uint64)
    err = OrcErr()
    if avx2:
        ptr, n = C.c_void_p(), C.c_size_t(0)
        st = lib().orc_avx2_decode(_p(e), e.size, n_bases, C.byref(ptr), C.byref(n), C.byref(err))
        out = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint8)
        lib().orc_free(ptr)
    else:
        out = np.zeros(n_bases, dtype=np.uint8)
        st = lib().orc_decode(_p(e), e.size, n_bases, _p(out), C.byref(err))
    if st:
        raise OracleError(err)
    return out


def hdist_scalar(u, v, length):
    out, err = C.c_uint32(0), OrcErr()
    if lib().orc_hdist_scalar(C.c_uint64(u), C.c_uint64(v), length, C.byref(out), C.byref(err)):
        raise OracleError(err)
    return out.value


def hdist(a, b, n_bases):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out, err = C.c_uint32(0), OrcErr()
    if lib().orc_hdist(_p(a), a.size, _p(b), b.size, n_bases, C.byref(out), C.byref(err)):
        raise OracleError(err)
    return out.value


def as_2bit_batch(kmers, k, stride, count):
    s = _u8(kmers)
    out, err = np.zeros(count, dtype=np.uint64), OrcErr()
    if lib().orc_as_2bit_batch(_p(s), k, stride, count, _p(out), C.byref(err)):
        raise OracleError(err)
    return out


def kmer_hdist_scan(ref, k, query):
    s = _u8(ref)
    nwin = s.size - k + 1 if (s.size >= k and k > 0) else 0
    out, err = np.zeros(nwin, dtype=np.uint8), OrcErr()
    if lib().orc_kmer_hdist_scan(_p(s), s.size, k, C.c_uint64(query), _p(out), C.byref(err)):
        raise OracleError(err)
    return out


def base_counts(words, n_bases):
    w = np.ascontiguousarray(words, dtype=np.uint64)
    out, err = np.zeros(4, dtype=np.uint64), OrcErr()
    if lib().orc_base_counts(_p(w), w.size, n_bases, _p(out), C.byref(err)):
        raise OracleError(err)
    return [int(x) for x in out]


def gc_content(words, n_bases):
    w = np.ascontiguousarray(words, dtype=np.uint64)
    return float(lib().orc_gc_content(_p(w), w.size, n_bases))


def hdist_pairs(a, b, length):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out, err = np.zeros(a.size, dtype=np.uint8), OrcErr()
    if lib().orc_hdist_pairs(_p(a), _p(b), a.size, length, _p(out), C.byref(err)):
        raise OracleError(err)
    return out


def split_packed(ebuf, slen, idx):
    """-> (lbuf, rbuf) as the reference leaves them (functions/split.rs:15-99, as written)."""
    e = np.ascontiguousarray(ebuf, dtype=np.uint64)
    lo, ro = np.zeros(e.size + 1, dtype=np.uint64), np.zeros(e.size + 1, dtype=np.uint64)
    nl, nr, err = C.c_size_t(0), C.c_size_t(0), OrcErr()
    if lib().orc_split_packed(_p(e), e.size, slen, idx, _p(lo), C.byref(nl), _p(ro), C.byref(nr), C.byref(err)):
        raise OracleError(err)
    return lo[: nl.value].copy(), ro[: nr.value].copy()


def nucgen(length, seed, first=0, flags=0):
    out = np.zeros(length, dtype=np.uint8)
    lib().orc_nucgen(_p(out), length, C.c_uint64(seed), C.c_uint64(first), flags)
    return out


def build_flags():
    """The compiler flags the oracle was built with (bench.py records them beside cpu_baseline)."""
    for line in open(os.path.join(ORACLE_DIR, "Makefile")):
        if line.startswith("CFLAGS"):
            return (os.environ.get("CC", "gcc") + " " + line.split("=", 1)[1].strip())
    return None


_native = None


def native_lib():
    """The same two oracle sources compiled `-O3 -march=native` ON THIS HOST into a temporary directory (never into the tree: the
    tracked build is -march=x86-64-v3 because it is built in the CPU container and travels).  BASELINE.md section 3 and the reference's
    .cargo/config.toml:1-2 (target-cpu=native) ask for the native figure; bench.py's cpu_baseline reports both.
    -> (CDLL, flags string) or (None, reason)."""
    global _native
    if _native is None:
        import tempfile
        d = tempfile.mkdtemp(prefix="bitnuc_oracle_native_")
        out = os.path.join(d, "libbitnuc_oracle_native.so")
        flags = ["-O3", "-march=native", "-fPIC", "-Wall", "-Wextra", "-std=c11", "-D_POSIX_C_SOURCE=200809L"]
        cc = os.environ.get("CC", "gcc")
        try:
            r = subprocess.run([cc, *flags, "-shared", "-o", out, os.path.join(ORACLE_DIR, "bitnuc_oracle.c"), os.path.join(ORACLE_DIR, "bitnuc_avx2.c")],
                               capture_output=True, text=True, timeout=300)
        except Exception as e:  # noqa: BLE001
            _native = (None, repr(e)[:200])
            return _native
        if r.returncode != 0:
            _native = (None, r.stderr[-300:])
            return _native
        L = C.CDLL(out)
        L.orc_avx2_time_roundtrip.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _native = (L, cc + " " + " ".join(flags[:2]) + " (built on this host into a temporary directory)")
    return _native


def avx2_time_roundtrip(seq, L=None):
    s = _u8(seq)
    e, d = C.c_double(0), C.c_double(0)
    st = (L or lib()).orc_avx2_time_roundtrip(_p(s), s.size, C.byref(e), C.byref(d))
    if st:
        raise RuntimeError(f"avx2 round trip failed: {st}")
    return e.value, d.value
