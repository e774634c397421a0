//! `extern "C"` declarations for include/bitnuc_hip.h (one per exported symbol).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_uint, c_void};

pub const BITNUC_OK: c_int = 0;
pub const BITNUC_INVALID_BASE: c_int = 1;
pub const BITNUC_SEQUENCE_TOO_LONG: c_int = 2;
pub const BITNUC_INVALID_LENGTH: c_int = 3;
pub const BITNUC_INDEX_OUT_OF_BOUNDS: c_int = 4;
pub const BITNUC_INVALID_RANGE: c_int = 5;
pub const BITNUC_UNSUPPORTED: c_int = 6;
pub const BITNUC_SPLIT_AS_WRITTEN: c_int = 0;
pub const BITNUC_SPLIT_CANONICAL: c_int = 1;
pub const BITNUC_BACKEND_ERROR: c_int = 100;

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct bitnuc_err {
    pub status: i32,
    pub backend_code: i32,
    pub value: u64,
    pub index: u64,
    pub byte: u8,
    pub _pad: [u8; 7],
}

#[repr(C)]
pub struct bitnuc_ctx {
    _private: [u8; 0],
}

#[repr(C)]
pub struct bitnuc_batch_plan {
    _private: [u8; 0],
}

#[repr(C)]
pub struct bitnuc_comm {
    _private: [u8; 0],
}
pub const BITNUC_UNIQUE_ID_BYTES: usize = 128;

extern "C" {
    pub fn bitnuc_version() -> *const c_char;
    pub fn bitnuc_ctx_create(device: c_int, out: *mut *mut bitnuc_ctx, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_ctx_create_on_stream(device: c_int, hip_stream: *mut c_void, out: *mut *mut bitnuc_ctx, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_ctx_destroy(ctx: *mut bitnuc_ctx);
    pub fn bitnuc_ctx_sync(ctx: *mut bitnuc_ctx, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_ctx_stream(ctx: *mut bitnuc_ctx) -> *mut c_void;
    pub fn bitnuc_ctx_set_variant(ctx: *mut bitnuc_ctx, key: *const c_char, value: c_int) -> c_int;

    pub fn bitnuc_as_2bit(ctx: *mut bitnuc_ctx, seq: *const u8, len: usize, out: *mut u64, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_from_2bit(ctx: *mut bitnuc_ctx, packed: u64, n: usize, out: *mut u8, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_hdist_scalar(ctx: *mut bitnuc_ctx, u: u64, v: u64, len: usize, out: *mut u32, err: *mut bitnuc_err) -> c_int;

    pub fn bitnuc_encode(ctx: *mut bitnuc_ctx, seq: *const u8, len: usize, out: *mut u64, n_words: *mut usize, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_decode(ctx: *mut bitnuc_ctx, ebuf: *const u64, n_words: usize, n_bases: usize, out: *mut u8, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_hdist(ctx: *mut bitnuc_ctx, a: *const u64, na: usize, b: *const u64, nb: usize, n_bases: usize, out: *mut u32, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_split_packed_sizes(n_words: usize, slen: usize, idx: usize, flags: c_int, n_left: *mut usize, n_right: *mut usize, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_split_packed(ctx: *mut bitnuc_ctx, ebuf: *const u64, n_words: usize, slen: usize, idx: usize, flags: c_int, lbuf: *mut u64, n_left: *mut usize, rbuf: *mut u64, n_right: *mut usize, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_split_packed_dev(ctx: *mut bitnuc_ctx, d_ebuf: *const u64, n_words: usize, slen: usize, idx: usize, flags: c_int, d_lbuf: *mut u64, d_rbuf: *mut u64, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_as_2bit_batch(ctx: *mut bitnuc_ctx, kmers: *const u8, k: usize, stride: usize, count: usize, out: *mut u64, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_kmer_hdist_scan(ctx: *mut bitnuc_ctx, reference: *const u8, n: usize, k: usize, query: u64, dist: *mut u8, err: *mut bitnuc_err) -> c_int;

    pub fn bitnuc_encode_dev(ctx: *mut bitnuc_ctx, d_seq: *const u8, len: usize, d_out: *mut u64, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_decode_dev(ctx: *mut bitnuc_ctx, d_ebuf: *const u64, n_words: usize, n_bases: usize, d_out: *mut u8, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_as_2bit_batch_dev(ctx: *mut bitnuc_ctx, d_kmers: *const u8, k: usize, stride: usize, count: usize, d_out: *mut u64, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_kmer_hdist_scan_dev(ctx: *mut bitnuc_ctx, d_ref: *const u8, n: usize, k: usize, query: u64, d_dist: *mut u8, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_hdist_dev(ctx: *mut bitnuc_ctx, d_a: *const u64, na: usize, d_b: *const u64, nb: usize, n_bases: usize, d_result: *mut u32, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_batch_word_offsets_dev(ctx: *mut bitnuc_ctx, d_offsets: *const u64, count: usize, d_word_offsets: *mut u64, total_words: *mut usize, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_encode_batch_dev(ctx: *mut bitnuc_ctx, d_seq: *const u8, d_offsets: *const u64, d_word_offsets: *const u64, count: usize, total_words: usize, d_out: *mut u64, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_decode_batch_dev(ctx: *mut bitnuc_ctx, d_words: *const u64, d_word_offsets: *const u64, d_offsets: *const u64, count: usize, total_words: usize, d_out: *mut u8, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_encode_batch(ctx: *mut bitnuc_ctx, seq: *const u8, offsets: *const u64, count: usize, out: *mut u64, out_cap_words: usize, word_offsets: *mut u64, n_words: *mut usize, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_decode_batch(ctx: *mut bitnuc_ctx, words: *const u64, word_offsets: *const u64, offsets: *const u64, count: usize, out: *mut u8, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_encode_fixed(ctx: *mut bitnuc_ctx, seq: *const u8, read_len: usize, stride: usize, count: usize, out: *mut u64, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_decode_fixed(ctx: *mut bitnuc_ctx, words: *const u64, read_len: usize, stride: usize, count: usize, out: *mut u8, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_encode_fixed_dev(ctx: *mut bitnuc_ctx, d_seq: *const u8, read_len: usize, stride: usize, count: usize, d_out: *mut u64, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_decode_fixed_dev(ctx: *mut bitnuc_ctx, d_words: *const u64, read_len: usize, stride: usize, count: usize, d_out: *mut u8, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_base_counts(ctx: *mut bitnuc_ctx, words: *const u64, n_words: usize, n_bases: usize, counts: *mut u64, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_base_counts_dev(ctx: *mut bitnuc_ctx, d_words: *const u64, n_words: usize, n_bases: usize, d_counts: *mut u64, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_hdist_pairs(ctx: *mut bitnuc_ctx, a: *const u64, b: *const u64, count: usize, len: usize, dist: *mut u8, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_hdist_query(ctx: *mut bitnuc_ctx, query: u64, targets: *const u64, count: usize, len: usize, dist: *mut u8, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_hdist_pairs_dev(ctx: *mut bitnuc_ctx, d_a: *const u64, d_b: *const u64, count: usize, len: usize, d_dist: *mut u8, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_hdist_query_dev(ctx: *mut bitnuc_ctx, query: u64, d_targets: *const u64, count: usize, len: usize, d_dist: *mut u8, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_comm_get_unique_id(id: *mut u8, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_comm_init_rank(ctx: *mut bitnuc_ctx, nranks: c_int, rank: c_int, id: *const u8, out: *mut *mut bitnuc_comm, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_comm_init_all(n_gpus: c_int, ctxs: *mut *mut bitnuc_ctx, comms: *mut *mut bitnuc_comm, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_comm_init_all_devices(n_gpus: c_int, devices: *const c_int, ctxs: *mut *mut bitnuc_ctx, comms: *mut *mut bitnuc_comm, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_comm_destroy(comm: *mut bitnuc_comm);
    pub fn bitnuc_comm_nranks(comm: *const bitnuc_comm) -> c_int;
    pub fn bitnuc_comm_rank(comm: *const bitnuc_comm) -> c_int;
    pub fn bitnuc_comm_single_process(comm: *const bitnuc_comm) -> c_int;
    pub fn bitnuc_allgather_words_dev(ctx: *mut bitnuc_ctx, comm: *mut bitnuc_comm, d_local: *const u64, count: usize, d_all: *mut u64, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_encode_sharded_allgather_dev(ctx: *mut bitnuc_ctx, comm: *mut bitnuc_comm, d_seq_shard: *const u8, shard_len: usize, d_all: *mut u64, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_encode_sharded_allgather_overlapped_dev(ctx: *mut bitnuc_ctx, comm: *mut bitnuc_comm, d_seq_shard: *const u8, shard_len: usize, n_chunks: c_int, d_all: *mut u64, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_peer_link_probe(src_device: c_int, dst_devices: *const c_int, n: c_int, bytes: usize, reps: c_int, gb_s_each: *mut f64, gb_s_all: *mut f64, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_host_pipe_info(ctx: *mut bitnuc_ctx, out: *mut f64, n: c_int, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_encode_sharded_allgather_all(n_gpus: c_int, ctxs: *mut *mut bitnuc_ctx, comms: *mut *mut bitnuc_comm, d_seq_shards: *const *const u8, shard_len: usize, d_alls: *const *mut u64, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_encode_sharded_allgather_overlapped_all(n_gpus: c_int, ctxs: *mut *mut bitnuc_ctx, comms: *mut *mut bitnuc_comm, d_seq_shards: *const *const u8, shard_len: usize, n_chunks: c_int, d_alls: *const *mut u64, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_batch_shard_ranges(offsets: *const u64, count: usize, nranks: c_int, seq_first: *mut usize, word_first: *mut u64, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_allgatherv_words_dev(ctx: *mut bitnuc_ctx, comm: *mut bitnuc_comm, counts: *const usize, d_all: *mut u64, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_allgatherv_words_all(n_gpus: c_int, ctxs: *mut *mut bitnuc_ctx, comms: *mut *mut bitnuc_comm, counts: *const usize, d_alls: *const *mut u64, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_comm_set_threaded(comm: *mut bitnuc_comm, threaded: c_int) -> c_int;
    pub fn bitnuc_nucgen_dev(ctx: *mut bitnuc_ctx, d_out: *mut u8, len: usize, seed: u64, first: u64, flags: c_int, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_stream_probe_dev(ctx: *mut bitnuc_ctx, mode: c_int, d_src: *const c_void, d_dst: *mut c_void, bytes: usize, err: *mut bitnuc_err) -> c_int;
    // layout plan of a ragged batch (include/bitnuc_hip.h): built once per offsets table, used by every encode / decode of it
    pub fn bitnuc_batch_plan_create(ctx: *mut bitnuc_ctx, out: *mut *mut bitnuc_batch_plan, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_batch_plan_build_dev(ctx: *mut bitnuc_ctx, plan: *mut bitnuc_batch_plan, d_offsets: *const u64, count: usize, total_words: *mut usize, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_batch_plan_destroy(plan: *mut bitnuc_batch_plan);
    pub fn bitnuc_batch_plan_total_words(plan: *const bitnuc_batch_plan) -> usize;
    pub fn bitnuc_batch_plan_count(plan: *const bitnuc_batch_plan) -> usize;
    pub fn bitnuc_batch_plan_word_offsets_dev(plan: *const bitnuc_batch_plan) -> *const u64;
    pub fn bitnuc_encode_batch_plan_dev(ctx: *mut bitnuc_ctx, plan: *const bitnuc_batch_plan, d_seq: *const u8, d_out: *mut u64, err: *mut bitnuc_err) -> c_int;
    pub fn bitnuc_decode_batch_plan_dev(ctx: *mut bitnuc_ctx, plan: *const bitnuc_batch_plan, d_words: *const u64, d_out: *mut u8, err: *mut bitnuc_err) -> c_int;
    // fused scan threshold (SURVEY 8d cfg 5): number of windows with distance <= tau
    pub fn bitnuc_kmer_hdist_count_dev(ctx: *mut bitnuc_ctx, d_ref: *const u8, n: usize, k: usize, query: u64, tau: c_uint, d_count: *mut u64, err: *mut bitnuc_err) -> c_int;
    // diagnostics
    pub fn bitnuc_selftime_small(op: c_int, n: usize, iters: usize) -> f64;
    pub fn bitnuc_selftime_host_copy(bytes: usize, threads: c_int, mode: c_int) -> f64;
}
