/*
 * bitnuc_hip.h -- C ABI of libbitnuc_hip.so: the MI355X (gfx950) replacement for
 * bitnuc's 2-bit pack/unpack + bulk encode/decode hot path.
 *
 * The reference (drbh/bitnuc, a pure-Rust crate) has no FFI of its own; the
 * drop-in boundary is its public Rust API (src/lib.rs:214-220).  Each entry point
 * below names the reference function it replaces.  A Rust `extern "C"` binding
 * for every symbol is shown in INTEGRATION.md / rust/src/ffi.rs; include/bitnuc.hpp
 * is the compiled C++ host layer with the reference's names and Vec semantics.
 *
 * All bulk compute happens in hand-written HIP kernels (bitnuc_amd/csrc/).  There is
 * no CPU fallback: without a usable HIP device context creation fails and every
 * device-pointer call and every bulk call at or above the host cutoff returns
 * BITNUC_BACKEND_ERROR / BITNUC_UNSUPPORTED.
 *
 * Size dispatch: the reference's as_2bit / from_2bit / hdist_scalar are
 * #[inline(always)] nanosecond functions, and a kernel launch with its copies costs
 * ~35 us, so the three single-word entry points and bulk HOST-POINTER calls below
 * `host_cutoff` bases (default: the measured host / GPU crossover, 1 Mi bases for encode and hdist, 512 Ki for decode; bitnuc_ctx_set_variant(ctx, "host_cutoff", n) or
 * BITNUC_HOST_CUTOFF) run as the library's own SWAR host code (csrc/host_word.h) and
 * accept ctx == NULL.  bitnuc_ctx_set_variant(ctx, "force_gpu", 1) or BITNUC_FORCE_GPU=1
 * sends every call MADE WITH THAT CONTEXT to the kernels (batches of one) -- the GPU parity
 * tests run that way; calls with ctx == NULL have no context to carry the flag and stay on the host.
 *
 * Asynchronous launches and their data errors.  Every _dev launch that can meet an invalid base owns
 * one device-resident error slot; bitnuc_ctx_sync() waits for the stream and reports the first latched
 * error: ordinary launches in launch order, then launches recorded into a hipGraph in capture order.
 *   - The slot ring starts at 4096 launches and GROWS (device + pinned allocation, no stream
 *     synchronisation) when more are queued between two syncs; only past 2 Mi unsynced launches does a
 *     _dev call drain the stream itself (the error it finds is kept for the next bitnuc_ctx_sync).
 *   - While the context's stream is being CAPTURED (hipStreamBeginCapture / torch.cuda.graph) a launch
 *     gets a persistent slot instead: the captured kernel runs again at every replay, so every later
 *     bitnuc_ctx_sync() examines that slot and re-arms it.  An InvalidBase met by a replay is therefore
 *     reported by the first sync after it, and never against another launch.  A context holds at most
 *     1024 captured launches over its life (BITNUC_UNSUPPORTED, err.value = 1024 beyond that); replay the
 *     graph on the context's stream, or order it before the sync yourself.  Entry points documented as
 *     synchronous (word offsets, plan build, every host-pointer call, bitnuc_ctx_sync) cannot be captured.
 *   - The table-driven batch calls (bitnuc_encode_batch_dev / bitnuc_decode_batch_dev) keep their layout plan in context scratch,
 *     sized by the largest batch seen.  A call that would have to GROW it while the stream is being captured returns
 *     BITNUC_UNSUPPORTED (err.value = the bytes needed) before anything is touched -- warm up with the largest batch, or use a
 *     bitnuc_batch_plan; a scratch buffer that a recorded launch was handed stays alive until bitnuc_ctx_destroy (a later,
 *     larger ordinary call allocates a new one), so replays never write into freed memory.
 *   - Data errors found by implicit drains (a host-pointer call starts from an empty slot ring, and so does the 2 Mi limit above) are
 *     kept in launch order, oldest first, one per bitnuc_ctx_sync: with errors A and B kept from two such drains and a third one, C,
 *     that only the sync's own drain finds, three syncs report A, B, C (at most 64 are kept between syncs).
 *
 * Conventions
 *   - plain pointers + sizes, no torch / C++ types in signatures;
 *   - return value == err->status (err may be NULL);
 *   - "host" entry points take host pointers and are synchronous; large encode / decode
 *     calls (>= 8 Mi bases) are pipelined in 32 Mi-base chunks: the calling thread copies chunk c from the
 *     caller's memory to the device and launches on it while ONE helper thread of the context copies chunk
 *     c-1's output into the caller's memory (the runtime pins pageable buffers in place: both copies run at
 *     the DMA rate, each blocks only the thread that issued it).  BITNUC_PIPE_IMPL=staged selects the earlier
 *     engine instead (the library's own pinned buffers, filled and emptied by 8 + 4 copy threads, capped by the
 *     CPUs this process may use -- affinity AND cgroup quota; BITNUC_HOST_THREADS / BITNUC_HOST_THREADS_LIGHT
 *     override); bitnuc_host_pipe_info reports the engine and that budget;
 *     bitnuc_as_2bit_batch, bitnuc_kmer_hdist_scan, bitnuc_encode_fixed and (back-to-back reads) bitnuc_decode_fixed
 *     ride the same engine from 8 MiB of input; the other host entry points (and smaller inputs) stage through
 *     device scratch in 128 Mbase chunks;
 *     "_dev" entry points take device pointers, are enqueued
 *     on the context's stream and return immediately -- data-dependent errors
 *     (InvalidBase) are latched on the device and reported by bitnuc_ctx_sync();
 *   - a bitnuc_ctx owns one device + stream + scratch and must not be used from
 *     two threads at once; separate contexts are independent.
 */
#ifndef BITNUC_HIP_H
#define BITNUC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* NucleotideError, src/error.rs:3-18, plus a backend status that has no reference
 * counterpart (HIP runtime failure). */
typedef enum bitnuc_status {
    BITNUC_OK = 0,
    BITNUC_INVALID_BASE = 1,        /* InvalidBase(u8)         -> err.byte, err.index */
    BITNUC_SEQUENCE_TOO_LONG = 2,   /* SequenceTooLong(usize)  -> err.value */
    BITNUC_INVALID_LENGTH = 3,      /* InvalidLength(usize)    -> err.value */
    BITNUC_INDEX_OUT_OF_BOUNDS = 4, /* IndexOutOfBounds{index,length} -> err.index, err.value (split_packed) */
    BITNUC_INVALID_RANGE = 5,       /* decreasing offsets in the ragged-batch entry points -> err.value = index */
    BITNUC_UNSUPPORTED = 6,         /* Unsupported (bad argument combination) */
    BITNUC_BACKEND_ERROR = 100      /* hipError_t in err.backend_code */
} bitnuc_status;

typedef struct bitnuc_err {
    int32_t status;       /* bitnuc_status */
    int32_t backend_code; /* hipError_t when status == BITNUC_BACKEND_ERROR */
    uint64_t value;       /* offending length for TOO_LONG / INVALID_LENGTH */
    uint64_t index;       /* absolute byte index of the first invalid base */
    uint8_t byte;         /* the first invalid base */
    uint8_t _pad[7];
} bitnuc_err;

typedef struct bitnuc_ctx bitnuc_ctx;

/* ---- context ------------------------------------------------------------------ */
/* Create a context on HIP device `device` with its own non-blocking stream. */
int bitnuc_ctx_create(int device, bitnuc_ctx **out, bitnuc_err *err);
/* Same, but enqueue on a caller-owned hipStream_t (e.g. torch's current stream;
 * NULL = the default stream).  The stream is not destroyed with the context. */
int bitnuc_ctx_create_on_stream(int device, void *hip_stream, bitnuc_ctx **out, bitnuc_err *err);
void bitnuc_ctx_destroy(bitnuc_ctx *ctx);
/* Wait for everything enqueued on the context's stream, then report (and clear)
 * the first latched data error of the launches since the last sync. */
int bitnuc_ctx_sync(bitnuc_ctx *ctx, bitnuc_err *err);
/* The context's hipStream_t, for callers that order their own work against it. */
void *bitnuc_ctx_stream(bitnuc_ctx *ctx);
/* Library / build identification: "bitnuc_hip <ver> gfx950 csrc:<16 hex digits>[ sweep]" -- the hex digits are the SHA-256 prefix of the
 * sources the library was compiled from (csrc/ incl. csrc/evidence/, this header), passed by bitnuc_amd/build.py as -DBITNUC_CSRC_SHA
 * ("csrc:unknown" for a build without it); " sweep" marks the evidence build (-DBITNUC_SWEEP_VARIANTS). */
const char *bitnuc_version(void);

/* ---- single-word API (host code, ctx may be NULL; see "Size dispatch" above) ------------------ */
/* as_2bit(seq:&[u8]) -> Result<u64>          src/utils/packing/mod.rs:80-110
 * len > 32 -> SEQUENCE_TOO_LONG(len) before any base is looked at; first bad byte
 * -> INVALID_BASE(byte, index); len == 0 -> OK, 0.  For many k-mers use bitnuc_as_2bit_batch. */
int bitnuc_as_2bit(bitnuc_ctx *ctx, const uint8_t *seq, size_t len, uint64_t *out, bitnuc_err *err);
/* from_2bit(packed, expected_size, &mut Vec<u8>)  src/utils/unpacking/mod.rs:119-147
 * n > 32 -> INVALID_LENGTH(n).  Writes exactly n bytes at out (the Vec append
 * offset is the caller's: include/bitnuc.hpp, rust/src/lib.rs). */
int bitnuc_from_2bit(bitnuc_ctx *ctx, uint64_t packed, size_t n, uint8_t *out, bitnuc_err *err);
/* hdist_scalar(u, v, len) -> Result<u32>     src/utils/functions/hamming/scalar.rs:11-48 */
int bitnuc_hdist_scalar(bitnuc_ctx *ctx, uint64_t u, uint64_t v, size_t len, uint32_t *out, bitnuc_err *err);

/* ---- bulk API, host pointers (synchronous) ------------------------------------ */
/* encode(sequence, &mut Vec<u64>)            src/utils/mod.rs:22-25 -> packing/avx.rs:130-151
 * out must hold ceil(len/32) words.  On success *n_words = ceil(len/32), last word
 * zero-padded high.  On INVALID_BASE: err.byte/err.index = first invalid byte of
 * the whole sequence and *n_words = err.index/32 (the words the reference's Vec
 * holds at that point); out[*n_words..] is unspecified.  len == 0 -> OK with 0
 * words (the reference panics there; shims may mirror that). */
int bitnuc_encode(bitnuc_ctx *ctx, const uint8_t *seq, size_t len, uint64_t *out, size_t *n_words, bitnuc_err *err);
/* decode(ebuf, n_bases, &mut Vec<u8>)        src/utils/mod.rs:60-62 -> unpacking/avx.rs:116-153
 * Writes n_bases bytes at out.  n_words < ceil(n_bases/32) -> INVALID_LENGTH(n_bases)
 * (the rule of unpacking/mod.rs:40-45; the AVX2 path panics there). */
int bitnuc_decode(bitnuc_ctx *ctx, const uint64_t *ebuf, size_t n_words, size_t n_bases, uint8_t *out, bitnuc_err *err);
/* hdist(ebuf1, ebuf2, n_bases) -> Result<u32>  src/utils/functions/hamming/multi.rs:121-160 */
int bitnuc_hdist(bitnuc_ctx *ctx, const uint64_t *a, size_t na, const uint64_t *b, size_t nb, size_t n_bases, uint32_t *out, bitnuc_err *err);
/* Batched as_2bit over `count` k-mers, k-mer j at kmers + j*stride (README.md:52-56
 * host-loop idiom; stride 1 = every window of a sequence, src/lib.rs:170-173).
 * k > 32 -> SEQUENCE_TOO_LONG(k).  First invalid examined byte -> INVALID_BASE.
 * Any stride >= 1 works; stride == k (dense) and every stride <= k below 32 (overlapping windows; 1 = all
 * windows of a sequence) have dedicated kernels and run about 3x faster than the general one. */
int bitnuc_as_2bit_batch(bitnuc_ctx *ctx, const uint8_t *kmers, size_t k, size_t stride, size_t count, uint64_t *out, bitnuc_err *err);
/* Sliding-window k-mer pack + Hamming distance to a packed query: dist[i] =
 * hdist_scalar(as_2bit(ref[i..i+k]), query, k) for i in 0..n-k+1
 * (composition of packing/mod.rs:80-110 and hamming/scalar.rs:11-48). */
int bitnuc_kmer_hdist_scan(bitnuc_ctx *ctx, const uint8_t *ref, size_t n, size_t k, uint64_t query, uint8_t *dist, bitnuc_err *err);

/* ---- bulk API, device pointers (asynchronous on the context's stream) ------------ */
/* The roofline entry points: inputs and outputs stay resident in HBM.  d_seq /
 * d_out may have any alignment (16-byte aligned input is the fast path).
 * Argument errors are returned immediately; InvalidBase is latched for
 * bitnuc_ctx_sync(). */
int bitnuc_encode_dev(bitnuc_ctx *ctx, const uint8_t *d_seq, size_t len, uint64_t *d_out, bitnuc_err *err);
int bitnuc_decode_dev(bitnuc_ctx *ctx, const uint64_t *d_ebuf, size_t n_words, size_t n_bases, uint8_t *d_out, bitnuc_err *err);
int bitnuc_as_2bit_batch_dev(bitnuc_ctx *ctx, const uint8_t *d_kmers, size_t k, size_t stride, size_t count, uint64_t *d_out, bitnuc_err *err);
int bitnuc_kmer_hdist_scan_dev(bitnuc_ctx *ctx, const uint8_t *d_ref, size_t n, size_t k, uint64_t query, uint8_t *d_dist, bitnuc_err *err);
/* The same scan with the fused threshold of SURVEY 8(d) cfg 5: *d_count (one uint64 in device memory) = number of windows i with
 * hdist_scalar(as_2bit(ref[i..i+k]), query, k) <= tau.  No distance bytes are written: 1 byte read per window. */
int bitnuc_kmer_hdist_count_dev(bitnuc_ctx *ctx, const uint8_t *d_ref, size_t n, size_t k, uint64_t query, unsigned tau, uint64_t *d_count, bitnuc_err *err);
/* d_result: one uint32 in device memory, overwritten with the distance. */
int bitnuc_hdist_dev(bitnuc_ctx *ctx, const uint64_t *d_a, size_t na, const uint64_t *d_b, size_t nb, size_t n_bases, uint32_t *d_result, bitnuc_err *err);

/* ---- ragged batches of independent sequences (reads, contigs) ----------------------------- */
/* The reference's idiom is a host loop `for s in seqs { encode(s, &mut ebuf)? }` /
 * `decode(&ebuf, len, &mut dbuf)?` (src/utils/mod.rs:22-25,60-62), each sequence padding its
 * own last word (packing/avx.rs:147-148).  Here the `count` sequences sit back to back:
 * sequence i = seq[offsets[i] .. offsets[i+1]) (offsets has count+1 non-decreasing entries)
 * and its ceil(len_i/32) words start at out[word_offsets[i]], where
 * word_offsets[i] = sum_{j<i} ceil(len_j/32) and word_offsets[count] = total words.
 * Zero-length sequences produce no words (the reference panics on them).  The first
 * invalid byte in buffer order -> INVALID_BASE{byte, index = byte offset in seq}.
 * Memory access: the batch and fixed-length kernels LOAD whole 16-byte aligned chunks, so up to
 * 15 bytes before seq[offsets[0]] and after seq[offsets[count]-1] are read (never validated, never
 * part of a word; an aligned 16-byte chunk cannot cross a page, so this cannot fault).  Stores
 * touch exactly the bytes of the batch: neighbours of an unaligned output are never written. */
/* Compute word_offsets (count+1 entries, device memory) from device offsets; synchronous;
 * *total_words = word_offsets[count]. */
int bitnuc_batch_word_offsets_dev(bitnuc_ctx *ctx, const uint64_t *d_offsets, size_t count, uint64_t *d_word_offsets, size_t *total_words, bitnuc_err *err);
int bitnuc_encode_batch_dev(bitnuc_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_offsets, const uint64_t *d_word_offsets, size_t count, size_t total_words, uint64_t *d_out, bitnuc_err *err);
/* Writes sequence i's bases at d_out[offsets[i] .. offsets[i+1]) (the layout encode_batch read). */
int bitnuc_decode_batch_dev(bitnuc_ctx *ctx, const uint64_t *d_words, const uint64_t *d_word_offsets, const uint64_t *d_offsets, size_t count, size_t total_words, uint8_t *d_out, bitnuc_err *err);
/* Host-pointer forms (synchronous).  encode: out must hold sum ceil(len_i/32) words
 * (<= (offsets[count]-offsets[0])/32 + count); word_offsets (count+1 entries) is an output.
 * offsets that decrease -> INVALID_RANGE{value = index of the offending entry}; decode: a word_offsets
 * table that is not the one encode_batch produces for these offsets -> INVALID_RANGE likewise (the
 * _dev forms trust their tables: they are the caller's device memory). */
int bitnuc_encode_batch(bitnuc_ctx *ctx, const uint8_t *seq, const uint64_t *offsets, size_t count, uint64_t *out, size_t out_cap_words, uint64_t *word_offsets, size_t *n_words, bitnuc_err *err);
int bitnuc_decode_batch(bitnuc_ctx *ctx, const uint64_t *words, const uint64_t *word_offsets, const uint64_t *offsets, size_t count, uint8_t *out, bitnuc_err *err);

/* Layout plan: everything about a ragged batch that depends only on its offsets table, computed once and used by
 * every encode / decode of that layout (a layout is used at least twice: encode now, decode later).  The plan holds
 * the word offsets, one byte offset per 64-word tile and one pad byte per word; with it a lane finds its word's bytes
 * with one byte load and a prefix sum -- no offsets window, no search, no per-call pre-kernel -- and the ragged
 * kernels run at the speed of the fixed-length ones.  _build_dev reads d_offsets (count+1 non-decreasing entries,
 * device memory) once, is synchronous and may be called again on the same plan for the next batch (memory is reused).
 * encode / decode take the buffers the table-driven entry points take: sequence i's bases at d_seq + offsets[i], its
 * words at d_out + word_offsets[i].  Same words, same InvalidBase rule (first invalid byte in buffer order). */
typedef struct bitnuc_batch_plan bitnuc_batch_plan;
int bitnuc_batch_plan_create(bitnuc_ctx *ctx, bitnuc_batch_plan **out, bitnuc_err *err);
int bitnuc_batch_plan_build_dev(bitnuc_ctx *ctx, bitnuc_batch_plan *plan, const uint64_t *d_offsets, size_t count, size_t *total_words, bitnuc_err *err);
void bitnuc_batch_plan_destroy(bitnuc_batch_plan *plan);
size_t bitnuc_batch_plan_total_words(const bitnuc_batch_plan *plan);
size_t bitnuc_batch_plan_count(const bitnuc_batch_plan *plan);
const uint64_t *bitnuc_batch_plan_word_offsets_dev(const bitnuc_batch_plan *plan); /* count+1 entries, device memory owned by the plan */
int bitnuc_encode_batch_plan_dev(bitnuc_ctx *ctx, const bitnuc_batch_plan *plan, const uint8_t *d_seq, uint64_t *d_out, bitnuc_err *err);
int bitnuc_decode_batch_plan_dev(bitnuc_ctx *ctx, const bitnuc_batch_plan *plan, const uint64_t *d_words, uint8_t *d_out, bitnuc_err *err);

/* Fixed-length reads (the common sequencing layout): `count` reads of `read_len` bases, read r
 * at byte r*stride (stride >= read_len; stride == read_len: back to back; stride == read_len+1:
 * newline-separated, ...).  Read r's ceil(read_len/32) words are out[r*wpr .. (r+1)*wpr), its
 * last word zero-padded high -- the same words as `encode(read_r)` per read.  Separator bytes
 * are never examined.  No offsets tables and no lookup: the fastest batch form. */
int bitnuc_encode_fixed_dev(bitnuc_ctx *ctx, const uint8_t *d_seq, size_t read_len, size_t stride, size_t count, uint64_t *d_out, bitnuc_err *err);
/* Inverse: read r's bases are written at d_out[r*stride .. r*stride + read_len); bytes between
 * reads are left untouched. */
int bitnuc_decode_fixed_dev(bitnuc_ctx *ctx, const uint64_t *d_words, size_t read_len, size_t stride, size_t count, uint8_t *d_out, bitnuc_err *err);
int bitnuc_encode_fixed(bitnuc_ctx *ctx, const uint8_t *seq, size_t read_len, size_t stride, size_t count, uint64_t *out, bitnuc_err *err);
int bitnuc_decode_fixed(bitnuc_ctx *ctx, const uint64_t *words, size_t read_len, size_t stride, size_t count, uint8_t *out, bitnuc_err *err);

/* ---- analysis on packed words (the callers just above the codec) ------ */
/* BaseCount::base_counts / GCContent::gc_content of a packed sequence
 * (src/utils/analysis.rs:7-39: the reference decodes to ASCII, then counts bytes): counts[] =
 * {A, C, G, T} of the first n_bases bases.  n_words < ceil(n_bases/32) -> INVALID_LENGTH.
 * gc_content = (counts[1] + counts[2]) as f64 / n_bases as f64 * 100.0 is left to the caller. */
int bitnuc_base_counts(bitnuc_ctx *ctx, const uint64_t *words, size_t n_words, size_t n_bases, uint64_t counts[4], bitnuc_err *err);
int bitnuc_base_counts_dev(bitnuc_ctx *ctx, const uint64_t *d_words, size_t n_words, size_t n_bases, uint64_t *d_counts /* 4, device */, bitnuc_err *err);
/* Many-pair / one-query Hamming distance of packed <=32-mers: dist[i] = hdist_scalar(a[i], b[i], len)
 * or hdist_scalar(query, targets[i], len) (src/utils/functions/hamming/scalar.rs:11-48).
 * len > 32 -> INVALID_LENGTH(len). */
int bitnuc_hdist_pairs_dev(bitnuc_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, size_t count, size_t len, uint8_t *d_dist, bitnuc_err *err);
int bitnuc_hdist_query_dev(bitnuc_ctx *ctx, uint64_t query, const uint64_t *d_targets, size_t count, size_t len, uint8_t *d_dist, bitnuc_err *err);
int bitnuc_hdist_pairs(bitnuc_ctx *ctx, const uint64_t *a, const uint64_t *b, size_t count, size_t len, uint8_t *dist, bitnuc_err *err);
int bitnuc_hdist_query(bitnuc_ctx *ctx, uint64_t query, const uint64_t *targets, size_t count, size_t len, uint8_t *dist, bitnuc_err *err);

/* ---- split_packed ----------------------------------------------------- */
/* split_packed(ebuf, slen, idx, &mut lbuf, &mut rbuf)   src/utils/functions/split.rs:15-99
 * Cut a packed sequence of slen bases at base idx (idx goes to the right part).
 * idx > slen -> INDEX_OUT_OF_BOUNDS{err.index = idx, err.value = slen} (split.rs:23-28).
 * flags:
 *   BITNUC_SPLIT_AS_WRITTEN  the reference's output, word for word: idx == 0 -> right = ebuf;
 *       idx == slen -> left = ebuf; otherwise left = ebuf[..idx/32] + the masked split word
 *       (idx/32 + 1 words, a zero word when idx % 32 == 0) and right = n_words - idx/32 words,
 *       right[j] = ebuf[idx/32 + j] >> s | ebuf[idx/32 + j - 1] << (64 - s), s = 2*(idx % 32)
 *       (split.rs:84-94 ORs in the low bits of the PREVIOUS word, so a right part longer than
 *       one word is not the shifted sequence unless s == 0; the reference's tests only cover
 *       one-word right parts and s == 0).
 *   BITNUC_SPLIT_CANONICAL   the funnel shift the function documents: left == encode(seq[..idx])
 *       (ceil(idx/32) words) and right == encode(seq[idx..]) (ceil((slen-idx)/32) words), bits
 *       above the last base cleared.  Agrees with AS_WRITTEN (up to trailing words / the zero
 *       word) wherever the reference's own tests look.
 * n_words < ceil(slen/32) -> INVALID_LENGTH(slen) (the reference panics or returns a
 * data-dependent length there).  lbuf / rbuf must not overlap ebuf.  _sizes returns the word
 * counts a call will write; the host form returns them too. */
#define BITNUC_SPLIT_AS_WRITTEN 0
#define BITNUC_SPLIT_CANONICAL 1
int bitnuc_split_packed_sizes(size_t n_words, size_t slen, size_t idx, int flags, size_t *n_left, size_t *n_right, bitnuc_err *err);
int bitnuc_split_packed(bitnuc_ctx *ctx, const uint64_t *ebuf, size_t n_words, size_t slen, size_t idx, int flags, uint64_t *lbuf, size_t *n_left, uint64_t *rbuf, size_t *n_right, bitnuc_err *err);
int bitnuc_split_packed_dev(bitnuc_ctx *ctx, const uint64_t *d_ebuf, size_t n_words, size_t slen, size_t idx, int flags, uint64_t *d_lbuf, uint64_t *d_rbuf, bitnuc_err *err);

/* ---- multi-GPU: shard + concatenate (BASELINE config 4) ------------------------------------ */
/* u64 words never share state (packing/avx.rs:138-145 has no carry), so a sequence split at
 * multiples of 32 bases encodes shard by shard with no communication; the only exchange is the
 * optional concatenation of the packed words, one RCCL all-gather over xGMI.  A communicator
 * rank is bound to one context (one device + stream).  librccl.so is loaded on first use.
 *   - one process (or thread) per GPU: rank 0 calls bitnuc_comm_get_unique_id, hands the 128 bytes to the
 *     other ranks (env, file, MPI, torch.distributed ...), every rank calls bitnuc_comm_init_rank and then
 *     the per-rank entry points (bitnuc_allgather_words_dev, bitnuc_encode_sharded_allgather_dev,
 *     bitnuc_encode_sharded_allgather_overlapped_dev), EACH RANK FROM ITS OWN THREAD OR PROCESS: like every
 *     collective they complete only when all ranks have called them, and RCCL may block the calling thread
 *     until its peers have;
 *   - one thread, all GPUs: bitnuc_comm_init_all[_devices] creates n contexts + communicators at once; that
 *     thread then drives all ranks with the _all entry points (bitnuc_encode_sharded_allgather_all,
 *     bitnuc_encode_sharded_allgather_overlapped_all), which issue every rank's part of an exchange inside one
 *     ncclGroupStart / ncclGroupEnd.  The per-rank entry points REFUSE such a communicator of more than one
 *     rank (BITNUC_UNSUPPORTED): called rank after rank from the one thread, the first call would wait for
 *     peers that thread has not reached yet.  bitnuc_comm_single_process() tells the two kinds apart. */
typedef struct bitnuc_comm bitnuc_comm;
#define BITNUC_UNIQUE_ID_BYTES 128
int bitnuc_comm_get_unique_id(uint8_t id[BITNUC_UNIQUE_ID_BYTES], bitnuc_err *err);
int bitnuc_comm_init_rank(bitnuc_ctx *ctx, int nranks, int rank, const uint8_t id[BITNUC_UNIQUE_ID_BYTES], bitnuc_comm **out, bitnuc_err *err);
int bitnuc_comm_init_all(int n_gpus, bitnuc_ctx **ctxs /* out: n_gpus */, bitnuc_comm **comms /* out: n_gpus */, bitnuc_err *err);
/* ... on the HIP devices devices[0..n_gpus) instead of 0..n_gpus-1 (rank i on devices[i]; NULL = the identity). */
int bitnuc_comm_init_all_devices(int n_gpus, const int *devices, bitnuc_ctx **ctxs, bitnuc_comm **comms, bitnuc_err *err);
void bitnuc_comm_destroy(bitnuc_comm *comm);
int bitnuc_comm_nranks(const bitnuc_comm *comm);
int bitnuc_comm_rank(const bitnuc_comm *comm);
/* 1: made by bitnuc_comm_init_all[_devices] (use the _all entry points), 0: by bitnuc_comm_init_rank, -1: NULL. */
int bitnuc_comm_single_process(const bitnuc_comm *comm);
/* All-gather `count` packed words per rank: d_all[r*count .. (r+1)*count) = rank r's d_local.
 * d_local may alias d_all + rank*count (in place).  Asynchronous on the context's stream. */
int bitnuc_allgather_words_dev(bitnuc_ctx *ctx, bitnuc_comm *comm, const uint64_t *d_local, size_t count, uint64_t *d_all, bitnuc_err *err);
/* Encode this rank's shard (shard_len bases, the same multiple of 32 on every rank) straight
 * into its slot of d_all and all-gather in place: every rank ends with the packed words of the
 * whole nranks*shard_len-base sequence, bit-identical to a single-GPU encode of it. */
int bitnuc_encode_sharded_allgather_dev(bitnuc_ctx *ctx, bitnuc_comm *comm, const uint8_t *d_seq_shard, size_t shard_len, uint64_t *d_all, bitnuc_err *err);
/* The same result with the exchange hidden behind the encode (SURVEY 8e iii): the shard is encoded in
 * n_chunks pieces on the context's stream while a second stream moves each finished piece IN PLACE
 * (grouped ncclSend / ncclRecv straight into d_all + peer*count + w0: no staging buffer, no strided copy;
 * BITNUC_GATHER_MODE=bcast uses grouped in-place ncclBroadcast instead).  The context's stream waits for
 * the exchange, so bitnuc_ctx_sync() covers the whole call; an InvalidBase index is relative to the shard. */
int bitnuc_encode_sharded_allgather_overlapped_dev(bitnuc_ctx *ctx, bitnuc_comm *comm, const uint8_t *d_seq_shard, size_t shard_len, int n_chunks, uint64_t *d_all, bitnuc_err *err);
/* Single-process forms over bitnuc_comm_init_all's contexts and communicators (all n of them, in rank order):
 * one call drives all n GPUs and synchronises every stream before it returns.  _all: encode on every device,
 * then one grouped ncclAllGather.  _overlapped_all: the chunked in-place exchange above -- per piece, the encode
 * on every device's stream, then ONE group holding every rank's sends and receives on that rank's transfer stream.
 * An InvalidBase is reported for the lowest rank that has one: err.index relative to that rank's shard,
 * err.value = the rank; the other ranks' slots are still exchanged. */
int bitnuc_encode_sharded_allgather_all(int n_gpus, bitnuc_ctx **ctxs, bitnuc_comm **comms, const uint8_t *const *d_seq_shards, size_t shard_len, uint64_t *const *d_alls, bitnuc_err *err);
int bitnuc_encode_sharded_allgather_overlapped_all(int n_gpus, bitnuc_ctx **ctxs, bitnuc_comm **comms, const uint8_t *const *d_seq_shards, size_t shard_len, int n_chunks, uint64_t *const *d_alls, bitnuc_err *err);

/* ---- a RAGGED BATCH across ranks ("for a batch of independent sequences of unequal length the partition is by whole
 * sequences with an offsets table; the final word of each sequence is padded independently" -- packing/avx.rs:147-148 is that
 * padding rule, src/utils/mod.rs:22-25 the per-sequence loop the batch replaces).
 * bitnuc_batch_shard_ranges: host arithmetic only, no device.  Sequence i holds bases [offsets[i], offsets[i+1]) and packs into
 * ceil(len / 32) words of its own.  Rank r of nranks gets the run of WHOLE sequences [seq_first[r], seq_first[r+1]), balanced by
 * word count: seq_first[r] = the first i whose word prefix W[i] >= floor(r * W[count] / nranks) (so a sequence longer than a fair
 * share stays whole and the ranks it covers get empty runs; seq_first[0] = 0, seq_first[nranks] = count).  word_first[r] =
 * W[seq_first[r]] = where rank r's words start in the concatenation; word_first[nranks] = the batch's total words.
 * Each rank then rebases its run's offsets to 0 and encodes it with the batch entry points above (bitnuc_batch_plan_build_dev +
 * bitnuc_encode_batch_plan_dev) straight into d_all + word_first[rank]; the global word_offsets table is word_first[rank] + the
 * rank's own table. */
int bitnuc_batch_shard_ranges(const uint64_t *offsets /* count+1 */, size_t count, int nranks, size_t *seq_first /* nranks+1 */, uint64_t *word_first /* nranks+1 */, bitnuc_err *err);
/* All-gather of UNEQUAL word counts, in place: counts[r] words of rank r live at d_all + sum(counts[0..r)) on rank r before the
 * call and on every rank after it (grouped ncclSend / ncclRecv with per-peer counts on the context's stream -- all P-1 links of a
 * GPU at once; BITNUC_GATHER_MODE=bcast: one grouped in-place ncclBroadcast per non-empty rank).  counts must be the same array
 * on every rank; a rank with counts[r] == 0 sends nothing.  Asynchronous like every _dev call. */
int bitnuc_allgatherv_words_dev(bitnuc_ctx *ctx, bitnuc_comm *comm, const size_t *counts /* nranks */, uint64_t *d_all, bitnuc_err *err);
/* ... for a thread that holds every rank (bitnuc_comm_init_all[_devices]): one group with every rank's sends and receives, then
 * every stream is synchronised (a data error latched by the encodes queued before it is reported like the other _all forms). */
int bitnuc_allgatherv_words_all(int n_gpus, bitnuc_ctx **ctxs, bitnuc_comm **comms, const size_t *counts /* n_gpus */, uint64_t *const *d_alls, bitnuc_err *err);
/* A communicator made by bitnuc_comm_init_all[_devices] is taken to be driven by ONE thread, and the per-rank entry points refuse
 * it (they would block in RCCL waiting for peers that thread has not issued).  A host that gives every rank of such a communicator
 * its own thread -- ordinary NCCL usage -- says so here (threaded = 1) and may then use the per-rank entry points; the _all forms
 * refuse the communicator from then on.  Returns the previous setting, -1 for NULL. */
int bitnuc_comm_set_threaded(bitnuc_comm *comm, int threaded);

/* xGMI link probe (no reference counterpart; SURVEY section 5 asks for the measured per-link rate before any
 * fabric fraction is quoted): hipMemcpyPeerAsync of `bytes` from src_device to each of dst_devices[0..n), one
 * link at a time (gb_s_each[i], best of reps) and then all n at once (*gb_s_all, the aggregate outbound rate).
 * All devices must be visible in this process and peer-accessible; fewer than two devices -> BITNUC_UNSUPPORTED
 * with err.value = the device count. */
int bitnuc_peer_link_probe(int src_device, const int *dst_devices, int n, size_t bytes, int reps, double *gb_s_each, double *gb_s_all, bitnuc_err *err);

/* ---- synthetic input (the reference's tests use nucgen::Sequence::fill_buffer,
 * src/utils/mod.rs:116-121; its stream is unpinned, so the build ships its own) ---- */
/* Fill d_out[0..len) with bases first..first+len of the seeded stream
 * base(i) = "ACGT"[(splitmix64(seed + (i/32+1)*0x9E3779B97F4A7C15) >> 2*(i%32)) & 3];
 * flags bit0: the benches' cyclic "ACGT"[i%4] pattern (benches/simd_comparison.rs:4-7);
 * bit1: lower-case mix, p = 0.25 (SURVEY 8d parity variant): base i is lower case iff 2-bit field i%32 of
 * splitmix64((seed ^ 0xC0FFEE5EEDC0DE55) + (i/32+1)*0x9E3779B97F4A7C15) is zero. */
int bitnuc_nucgen_dev(bitnuc_ctx *ctx, uint8_t *d_out, size_t len, uint64_t seed, uint64_t first, int flags, bitnuc_err *err);

/* ---- tuning / diagnostics (not part of the drop-in surface) ---------------------- */
/* Knobs: key in {"force_gpu", "host_cutoff", "host_pipeline"} (see "Size dispatch"), kernel-variant
 * selectors {"encode", "decode", "grid_mult", ...} for experiments.  A negative value only queries.
 * Returns the previous value, -1 for an unknown key, -2 for a variant this build does not hold (the
 * product library ships 4 of the 47 encode/decode variants; libbitnuc_hip_sweep.so, built with
 * -DBITNUC_SWEEP_VARIANTS, holds all of them plus the ballot formulation). */
int bitnuc_ctx_set_variant(bitnuc_ctx *ctx, const char *key, int value);
/* Pure streaming kernels used to measure the box's HBM ceiling next to the codec:
 * mode bits 0-2: 0 = read-only sum of `bytes` from d_src; 1 = copy d_src -> d_dst;
 * 2 = write-only fill of d_dst; 3 = encode_kernel's access shape (16 B nt load + 4 B store per lane, 2 in
 * flight, 128-thread workgroups) and 4 = decode_kernel's (4 B load + 16 B nt store, 256-thread workgroups),
 * both without arithmetic, `bytes` = ASCII-side bytes.  bit 3: nt loads, bit 4: nt stores, bit 5: 2 (not 4)
 * 16-byte groups in flight per lane (modes 0-2). */
int bitnuc_stream_probe_dev(bitnuc_ctx *ctx, int mode, const void *d_src, void *d_dst, size_t bytes, bitnuc_err *err);

/* Mean ns per call of the HOST path (op 0 as_2bit, 1 from_2bit, 2 encode, 3 decode, 4 hdist_scalar) on n bases of the
 * reference's bench input (benches/simd_comparison.rs:4-7), timed inside the library over `iters` calls; < 0 = bad argument. */
double bitnuc_selftime_small(int op, size_t n, size_t iters);
/* GB/s of the host-path staging pool's parallel memcpy (tools/host_path.py): mode 0 pageable -> pageable, 1 pageable -> pinned,
 * 2 pinned -> pageable; < 0 on failure. */
double bitnuc_selftime_host_copy(size_t bytes, int threads, int mode);

/* Configuration of this context's pipelined host-pointer path (creates it if needed): out[0..n) = cores_visible, cores_quota
 * (0 = none), cores_usable, chunk_bases, depth, encode stage-in / hand-back threads, decode stage-in / hand-back threads,
 * heavy_cap (the most threads the heavy side may use here), the GPU's NUMA node (-1 = unknown), the number of that node's CPUs
 * the staged engine's copy workers are bound to (0 = not bound; BITNUC_PIPE_NUMA=0 disables the binding), the engine in use
 * (1 = direct, 0 = staged). */
int bitnuc_host_pipe_info(bitnuc_ctx *ctx, double *out, int n, bitnuc_err *err);

#ifdef __cplusplus
}
#endif
#endif /* BITNUC_HIP_H */
