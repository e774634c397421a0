"""ctypes binding of libbitnuc_hip.so (the C ABI in include/bitnuc_hip.h).

There is no CPU fallback: if the shared library is missing this module raises
at import of the symbol table, and without a HIP device `Context()` raises `BackendError`
(every device-pointer call and every bulk call needs a context).  The three single-word
functions and host-pointer calls below the host cutoff are host code inside the same
library (include/bitnuc_hip.h, "Size dispatch").
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libbitnuc_hip.so")
SWEEP_LIB_PATH = os.path.join(HERE, "libbitnuc_hip_sweep.so")  # all 47 codec variants + the ballot formulation (evidence build)

# bitnuc_status (include/bitnuc_hip.h) == NucleotideError (src/error.rs:3-18)
OK, INVALID_BASE, SEQUENCE_TOO_LONG, INVALID_LENGTH = 0, 1, 2, 3
INDEX_OUT_OF_BOUNDS, INVALID_RANGE, UNSUPPORTED, BACKEND_ERROR = 4, 5, 6, 100
SPLIT_AS_WRITTEN, SPLIT_CANONICAL = 0, 1


class BitnucErr(C.Structure):
    _fields_ = [("status", C.c_int32), ("backend_code", C.c_int32), ("value", C.c_uint64),
                ("index", C.c_uint64), ("byte", C.c_uint8), ("_pad", C.c_uint8 * 7)]


_P = C.c_void_p
_SZ = C.c_size_t
_U64 = C.c_uint64
_ERR = C.POINTER(BitnucErr)

# name -> (restype, argtypes); every symbol include/bitnuc_hip.h declares
SIGNATURES = {
    "bitnuc_version": (C.c_char_p, []),
    "bitnuc_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P), _ERR]),
    "bitnuc_ctx_create_on_stream": (C.c_int, [C.c_int, _P, C.POINTER(_P), _ERR]),
    "bitnuc_ctx_destroy": (None, [_P]),
    "bitnuc_ctx_sync": (C.c_int, [_P, _ERR]),
    "bitnuc_ctx_stream": (_P, [_P]),
    "bitnuc_ctx_set_variant": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "bitnuc_as_2bit": (C.c_int, [_P, _P, _SZ, C.POINTER(_U64), _ERR]),
    "bitnuc_from_2bit": (C.c_int, [_P, _U64, _SZ, _P, _ERR]),
    "bitnuc_hdist_scalar": (C.c_int, [_P, _U64, _U64, _SZ, C.POINTER(C.c_uint32), _ERR]),
    "bitnuc_encode": (C.c_int, [_P, _P, _SZ, _P, C.POINTER(_SZ), _ERR]),
    "bitnuc_decode": (C.c_int, [_P, _P, _SZ, _SZ, _P, _ERR]),
    "bitnuc_hdist": (C.c_int, [_P, _P, _SZ, _P, _SZ, _SZ, C.POINTER(C.c_uint32), _ERR]),
    "bitnuc_as_2bit_batch": (C.c_int, [_P, _P, _SZ, _SZ, _SZ, _P, _ERR]),
    "bitnuc_kmer_hdist_scan": (C.c_int, [_P, _P, _SZ, _SZ, _U64, _P, _ERR]),
    "bitnuc_encode_dev": (C.c_int, [_P, _P, _SZ, _P, _ERR]),
    "bitnuc_decode_dev": (C.c_int, [_P, _P, _SZ, _SZ, _P, _ERR]),
    "bitnuc_as_2bit_batch_dev": (C.c_int, [_P, _P, _SZ, _SZ, _SZ, _P, _ERR]),
    "bitnuc_kmer_hdist_scan_dev": (C.c_int, [_P, _P, _SZ, _SZ, _U64, _P, _ERR]),
    "bitnuc_kmer_hdist_count_dev": (C.c_int, [_P, _P, _SZ, _SZ, _U64, C.c_uint, _P, _ERR]),
    "bitnuc_hdist_dev": (C.c_int, [_P, _P, _SZ, _P, _SZ, _SZ, _P, _ERR]),
    "bitnuc_batch_word_offsets_dev": (C.c_int, [_P, _P, _SZ, _P, C.POINTER(_SZ), _ERR]),
    "bitnuc_encode_batch_dev": (C.c_int, [_P, _P, _P, _P, _SZ, _SZ, _P, _ERR]),
    "bitnuc_decode_batch_dev": (C.c_int, [_P, _P, _P, _P, _SZ, _SZ, _P, _ERR]),
    "bitnuc_encode_batch": (C.c_int, [_P, _P, _P, _SZ, _P, _SZ, _P, C.POINTER(_SZ), _ERR]),
    "bitnuc_decode_batch": (C.c_int, [_P, _P, _P, _P, _SZ, _P, _ERR]),
    "bitnuc_batch_plan_create": (C.c_int, [_P, C.POINTER(_P), _ERR]),
    "bitnuc_batch_plan_build_dev": (C.c_int, [_P, _P, _P, _SZ, C.POINTER(_SZ), _ERR]),
    "bitnuc_batch_plan_destroy": (None, [_P]),
    "bitnuc_batch_plan_total_words": (_SZ, [_P]),
    "bitnuc_batch_plan_count": (_SZ, [_P]),
    "bitnuc_batch_plan_word_offsets_dev": (_P, [_P]),
    "bitnuc_encode_batch_plan_dev": (C.c_int, [_P, _P, _P, _P, _ERR]),
    "bitnuc_decode_batch_plan_dev": (C.c_int, [_P, _P, _P, _P, _ERR]),
    "bitnuc_encode_fixed_dev": (C.c_int, [_P, _P, _SZ, _SZ, _SZ, _P, _ERR]),
    "bitnuc_decode_fixed_dev": (C.c_int, [_P, _P, _SZ, _SZ, _SZ, _P, _ERR]),
    "bitnuc_encode_fixed": (C.c_int, [_P, _P, _SZ, _SZ, _SZ, _P, _ERR]),
    "bitnuc_decode_fixed": (C.c_int, [_P, _P, _SZ, _SZ, _SZ, _P, _ERR]),
    "bitnuc_base_counts": (C.c_int, [_P, _P, _SZ, _SZ, _P, _ERR]),
    "bitnuc_base_counts_dev": (C.c_int, [_P, _P, _SZ, _SZ, _P, _ERR]),
    "bitnuc_hdist_pairs_dev": (C.c_int, [_P, _P, _P, _SZ, _SZ, _P, _ERR]),
    "bitnuc_hdist_query_dev": (C.c_int, [_P, _U64, _P, _SZ, _SZ, _P, _ERR]),
    "bitnuc_hdist_pairs": (C.c_int, [_P, _P, _P, _SZ, _SZ, _P, _ERR]),
    "bitnuc_hdist_query": (C.c_int, [_P, _U64, _P, _SZ, _SZ, _P, _ERR]),
    "bitnuc_split_packed_sizes": (C.c_int, [_SZ, _SZ, _SZ, C.c_int, C.POINTER(_SZ), C.POINTER(_SZ), _ERR]),
    "bitnuc_split_packed": (C.c_int, [_P, _P, _SZ, _SZ, _SZ, C.c_int, _P, C.POINTER(_SZ), _P, C.POINTER(_SZ), _ERR]),
    "bitnuc_split_packed_dev": (C.c_int, [_P, _P, _SZ, _SZ, _SZ, C.c_int, _P, _P, _ERR]),
    "bitnuc_comm_get_unique_id": (C.c_int, [_P, _ERR]),
    "bitnuc_comm_init_rank": (C.c_int, [_P, C.c_int, C.c_int, _P, C.POINTER(_P), _ERR]),
    "bitnuc_comm_init_all": (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(_P), _ERR]),
    "bitnuc_comm_init_all_devices": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(_P), C.POINTER(_P), _ERR]),
    "bitnuc_comm_destroy": (None, [_P]),
    "bitnuc_comm_nranks": (C.c_int, [_P]),
    "bitnuc_comm_rank": (C.c_int, [_P]),
    "bitnuc_comm_single_process": (C.c_int, [_P]),
    "bitnuc_allgather_words_dev": (C.c_int, [_P, _P, _P, _SZ, _P, _ERR]),
    "bitnuc_encode_sharded_allgather_dev": (C.c_int, [_P, _P, _P, _SZ, _P, _ERR]),
    "bitnuc_encode_sharded_allgather_overlapped_dev": (C.c_int, [_P, _P, _P, _SZ, C.c_int, _P, _ERR]),
    "bitnuc_encode_sharded_allgather_all": (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), _SZ, C.POINTER(_P), _ERR]),
    "bitnuc_encode_sharded_allgather_overlapped_all": (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), _SZ, C.c_int, C.POINTER(_P), _ERR]),
    "bitnuc_batch_shard_ranges": (C.c_int, [_P, _SZ, C.c_int, _P, _P, _ERR]),
    "bitnuc_allgatherv_words_dev": (C.c_int, [_P, _P, _P, _P, _ERR]),
    "bitnuc_allgatherv_words_all": (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(_P), _P, C.POINTER(_P), _ERR]),
    "bitnuc_comm_set_threaded": (C.c_int, [_P, C.c_int]),
    "bitnuc_nucgen_dev": (C.c_int, [_P, _P, _SZ, _U64, _U64, C.c_int, _ERR]),
    "bitnuc_stream_probe_dev": (C.c_int, [_P, C.c_int, _P, _P, _SZ, _ERR]),
    "bitnuc_selftime_small": (C.c_double, [C.c_int, _SZ, _SZ]),
    "bitnuc_selftime_host_copy": (C.c_double, [_SZ, C.c_int, C.c_int]),
    "bitnuc_host_pipe_info": (C.c_int, [_P, C.POINTER(C.c_double), C.c_int, _ERR]),
    "bitnuc_peer_link_probe": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.c_int, _SZ, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), _ERR]),
}

_libs = {}


hip_runtime_choice = None  # what _share_torch_hip_runtime decided, for logs and tests: (reason, path or None)


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same soname as /opt/rocm's, requested under another name).
    If libbitnuc_hip.so pulls in the system runtime first, a later `import torch` loads a SECOND HIP runtime into the
    process and fails with "No HIP GPUs are available".  Loading torch's copy first makes both sides share one runtime
    (our DT_NEEDED libamdhip64.so.7 then resolves to it by soname) -- the order `import torch; import bitnuc_amd` gives.
    A plain C-ABI consumer that never imports torch can opt out with BITNUC_NO_TORCH_HIP_PRELOAD=1 and gets the runtime the
    library was linked against; BITNUC_LOG=1 prints the choice to stderr."""
    import importlib.util
    import sys
    global hip_runtime_choice

    def decided(reason, path=None):
        global hip_runtime_choice
        hip_runtime_choice = (reason, path)
        if os.environ.get("BITNUC_LOG"):
            print(f"bitnuc_amd: HIP runtime: {reason}" + (f" ({path})" if path else ""), file=sys.stderr)

    if os.environ.get("BITNUC_NO_TORCH_HIP_PRELOAD"):
        return decided("system runtime (BITNUC_NO_TORCH_HIP_PRELOAD set)")
    if "torch" in sys.modules:
        return decided("torch already imported: its runtime is in the process")
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return decided("system runtime (torch not installed)")
    if spec and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
                return decided("preloaded torch's bundled runtime", cand)
            except OSError as e:
                return decided(f"system runtime (preload of torch's runtime failed: {e})")
    return decided("system runtime (torch bundles no libamdhip64.so)")


def load(path=None):
    """Load libbitnuc_hip.so (or the library at `path`) and type every exported entry point."""
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -m bitnuc_amd.build` "
            "(hipcc, gfx950). bitnuc_amd has no CPU fallback.")
    if os.path.isdir(os.path.join(HERE, "csrc")) and not os.environ.get("BITNUC_ALLOW_STALE_LIB"):
        # the sources are beside the binary (a checkout, not an installed package): refuse a binary that was not built from them
        from . import build
        have, want = build.library_sha16(path), build.csrc_sha16()
        if have != want:
            raise RuntimeError(f"{path} was built from other sources (library csrc:{have}, sources csrc:{want}): rebuild with "
                               "`python -m bitnuc_amd.build` (bitnuc_amd.build.ensure_built() does it) or set BITNUC_ALLOW_STALE_LIB=1")
    _share_torch_hip_runtime()
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _libs[path] = lib
    return lib
