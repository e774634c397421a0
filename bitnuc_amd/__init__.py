"""bitnuc_amd -- MI355X (gfx950) implementation of bitnuc's 2-bit pack/unpack and
bulk encode/decode hot path, behind the C ABI of include/bitnuc_hip.h.

Layout
  csrc/     hand-written HIP kernels + the extern "C" runtime (libbitnuc_hip.so)
  _lib.py   ctypes binding of the C ABI
  api.py    host-side mirror of the reference's public API (src/lib.rs:214-220)
  dist.py   one-process-per-GPU sharding helpers (torch.distributed / RCCL)
"""
from .api import (BackendError, BatchPlan, Comm, CommGroup, Context, NucleotideError, as_2bit, as_2bit_batch, batch_shard_ranges, decode,
                  default_context, encode, encode_alloc, from_2bit, from_2bit_alloc, hdist,
                  hdist_scalar, kmer_hdist_scan, split_packed)

from .sequence import PackedSequence

__all__ = ["PackedSequence", "BatchPlan", "Comm", "CommGroup", "BackendError", "Context", "NucleotideError", "as_2bit", "as_2bit_batch", "batch_shard_ranges", "decode",
           "default_context", "encode", "encode_alloc", "from_2bit", "from_2bit_alloc", "hdist",
           "hdist_scalar", "kmer_hdist_scan", "split_packed"]
