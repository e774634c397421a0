// codec.hip -- bulk 2-bit encode / decode behind the C ABI (include/bitnuc_hip.h): the device-pointer entry points (the
// roofline path), the host-pointer calls with their size dispatch and the pipelined staging of large pageable buffers, the
// single-word API (host code, host_word.h), the synthetic generator and the streaming probes.  Kernels: codec_device.h.
// Reference: encode -> src/utils/mod.rs:22-25 -> packing/avx.rs:130-151; decode -> src/utils/mod.rs:60-62 ->
// unpacking/avx.rs:116-153; as_2bit / from_2bit -> packing/mod.rs:80-110, unpacking/mod.rs:119-147.
#include "runtime.h"
#include "codec_device.h"
#include "host_pipe.h"
#include "host_word.h"

#include <stdlib.h>
#include <time.h>

using namespace bitnuc_dev;
using namespace bitnuc_rt;
using bitnuc_host::CopyPool;

namespace {

// ---- kernel-variant tables -------------------------------------------------------
// The product library ships the variants that are in use: the tuned defaults (encode 39, decode 22), the plain
// reference shape (0) and an earlier default (3).  The other 43 and the lane-per-base ballot formulation are
// measurement evidence (profiles/): they are compiled only with -DBITNUC_SWEEP_VARIANTS, into
// libbitnuc_hip_sweep.so, which tools/sweep*.py and the all-variants parity test load.
//              id  UNROLL BLOCK NTLD   NTST   XPOSE  XCD
#ifdef BITNUC_SWEEP_VARIANTS
#define BITNUC_VARIANTS(X)                            \
    X(0, 4, 256, false, false, false, false)          \
    X(1, 4, 256, true, true, false, false)            \
    X(2, 2, 256, true, true, false, false)            \
    X(3, 2, 256, true, false, false, false)           \
    X(4, 2, 256, false, false, false, false)          \
    X(5, 4, 256, true, false, false, false)           \
    X(6, 1, 256, true, false, false, false)           \
    X(7, 2, 512, true, false, false, false)           \
    X(8, 2, 1024, true, false, false, false)          \
    X(9, 4, 256, false, false, true, false)           \
    X(10, 4, 256, true, true, true, false)            \
    X(11, 4, 256, true, false, true, false)           \
    X(12, 2, 256, true, false, false, true)           \
    X(13, 4, 256, false, false, false, true)          \
    X(14, 2, 128, true, false, false, false)          \
    X(15, 8, 256, true, false, false, false)          \
    X(16, 4, 512, false, false, false, false)         \
    X(17, 2, 512, false, false, false, false)         \
    X(18, 4, 256, false, true, false, false)          \
    X(19, 4, 512, true, false, true, false)           \
    X(20, 1, 256, false, false, false, false)         \
    X(21, 1, 512, true, false, false, false)          \
    X(22, 2, 256, false, true, false, false)          \
    X(23, 4, 1024, false, false, false, false)        \
    X(24, 2, 256, false, false, false, true)          \
    X(25, 4, 256, true, false, false, true)           \
    X(26, 8, 256, false, false, false, true)          \
    X(27, 4, 512, false, false, false, true)          \
    X(28, 4, 256, false, false, true, true)           \
    X(29, 1, 256, false, false, false, true)          \
    X(30, 2, 256, true, true, false, true)            \
    X(31, 4, 256, true, true, false, true)            \
    X(32, 8, 256, true, true, false, false)           \
    X(33, 4, 512, true, true, false, false)           \
    X(34, 4, 256, false, true, false, true)           \
    X(35, 4, 128, true, true, false, false)           \
    X(36, 2, 64, true, false, false, false)           \
    X(37, 4, 128, true, false, false, false)          \
    X(38, 1, 128, true, false, false, false)          \
    X(39, 2, 128, true, true, false, true)            \
    X(40, 4, 128, true, true, false, true)            \
    X(41, 2, 512, true, true, false, true)            \
    X(42, 1, 256, true, true, false, true)            \
    X(43, 2, 128, true, false, false, true)           \
    X(44, 1, 128, true, true, false, true)            \
    X(45, 1, 512, true, true, false, true)            \
    X(46, 1, 1024, true, true, false, true)
#else
#define BITNUC_VARIANTS(X)                            \
    X(0, 4, 256, false, false, false, false)          \
    X(3, 2, 256, true, false, false, false)           \
    X(22, 2, 256, false, true, false, false)          \
    X(39, 2, 128, true, true, false, true)
#endif
constexpr int kNumVariants = 47;    // ids 0..46; which of them this build holds: variant_info(id).built
[[maybe_unused]] constexpr int kBallotVariant = 100; // encode only: lane-per-base + ballot (sweep build; set_variant("encode", 100))
// defaults from the sustained (back-to-back) pair sweeps in profiles/ (10^9 bases, one tile per
// workgroup, interleaved rounds in one process, decode reading words written two steps
// earlier so that none of its input is Infinity-Cache resident -- what bench.py times):
//   encode 39: nt loads + nt stores, 2 groups in flight per lane, 128-thread workgroups, XCD-contiguous tile order
//   decode 22: plain loads + nt stores, 2 groups per lane
// The pair is tuned, not each kernel, and every good pair lands on the same plateau of ~0.40 ms per step = 6.3 TB/s of
// mixed read/write HBM traffic (all 47 x 47 pairs: profiles/r02_sweep_pairs_all_cold.txt; the ten best are within 0.6 %).
// What differs is how the step divides: with encode 14 (plain, allocating stores -- the round-1 default) the 250 MB of
// packed words sit dirty in the 256 MiB Infinity Cache and are written back while the DECODE runs: encode 0.182 ms, decode
// 0.214 ms.  With nt stores the encode pays for its own writes: 0.199-0.204 / 0.194 ms; the step is the same (five processes each:
// 0.4019 vs 0.4024 ms, profiles/r02_encode_variant_stability.txt): a choice of attribution, not of speed.
static_assert(kDefaultEnc == 39 && kDefaultDec == 22, "runtime.h holds the defaults the context starts with");

struct VariantInfo { int unroll, block; bool ntld, ntst, xpose, xcd, built; };
constexpr VariantInfo variant_info(int id) {
    switch (id) {
#define X(vid, U, B, NL, NS, XP, XC) case vid: return VariantInfo{U, B, NL, NS, XP, XC, true};
        BITNUC_VARIANTS(X)
#undef X
    default: return VariantInfo{0, 0, false, false, false, false, false};
    }
}

template <int UNROLL, int BLOCK, bool NTLD, bool NTST, bool XPOSE, bool XCD>
hipError_t launch_encode_t(bitnuc_ctx *c, const uint8_t *seq, uint32_t *out32, unsigned long long len,
                           unsigned long long *slot, bool al) {
    const unsigned long long tile = (unsigned long long)BLOCK * UNROLL;
    const unsigned grid = grid_for(c, (len >> 4) / tile + 1, BLOCK);
    // knobs(c).dyn_lds (evidence build, tools/ab_occupancy.py): unused dynamic LDS that only limits how many workgroups a CU holds
    if (al) encode_kernel<UNROLL, BLOCK, NTLD, NTST, true, XPOSE, XCD><<<grid, BLOCK, knobs(c).dyn_lds, c->stream>>>(seq, out32, len, slot);
    else if constexpr (!XPOSE) encode_kernel<UNROLL, BLOCK, NTLD, NTST, false, false, XCD><<<grid, BLOCK, knobs(c).dyn_lds, c->stream>>>(seq, out32, len, slot);
    return hipGetLastError();
}

constexpr int kQuadFirst = 47, kQuadLast = 62; // encode variants of the evidence build's encode_quad_kernel
#ifdef BITNUC_SWEEP_VARIANTS
// encode variants 47..62 (evidence build): encode_quad_kernel (16-byte stores by a register quad transpose, 4 rounds per wave).
// id - 47: bit 0 = nt loads, bit 1 = nt stores, bit 2 = XCD-contiguous tile order, bit 3 = 256 (not 128) threads per workgroup.
template <int BLOCK>
hipError_t launch_encode_quad_t(bitnuc_ctx *c, int mode, const uint8_t *seq, uint32_t *out32, unsigned long long len, unsigned long long *slot) {
    const unsigned grid = grid_for(c, (len >> 4) / ((unsigned long long)BLOCK * 4) + 1, BLOCK);
#define QUAD(NL, NS, XC) encode_quad_kernel<BLOCK, NL, NS, XC><<<grid, BLOCK, 0, c->stream>>>(seq, out32, len, slot)
    switch (mode & 7) {
    case 0: QUAD(false, false, false); break;
    case 1: QUAD(true, false, false); break;
    case 2: QUAD(false, true, false); break;
    case 3: QUAD(true, true, false); break;
    case 4: QUAD(false, false, true); break;
    case 5: QUAD(true, false, true); break;
    case 6: QUAD(false, true, true); break;
    default: QUAD(true, true, true); break;
    }
#undef QUAD
    return hipGetLastError();
}
#endif

hipError_t launch_encode(bitnuc_ctx *c, const uint8_t *seq, uint64_t *out, unsigned long long len, unsigned long long *slot) {
    uint32_t *o = reinterpret_cast<uint32_t *>(out);
    const bool in_al = aligned16(seq), out_al = aligned16(out);
#ifdef BITNUC_SWEEP_VARIANTS
    if (c->enc_variant >= kQuadFirst && c->enc_variant <= kQuadLast && in_al && out_al) {
        const int mode = c->enc_variant - kQuadFirst;
        return (mode & 8) ? launch_encode_quad_t<256>(c, mode, seq, o, len, slot) : launch_encode_quad_t<128>(c, mode, seq, o, len, slot);
    }
#endif
#ifdef BITNUC_SWEEP_VARIANTS
    if (c->enc_variant == kBallotVariant) { // lane-per-base + ballot formulation (evidence variant)
        const unsigned grid = grid_for(c, ((len + 63) / 64 + (kBlock / 64) * 4 - 1) / ((kBlock / 64) * 4));
        encode_ballot_kernel<4><<<grid, kBlock, 0, c->stream>>>(seq, reinterpret_cast<unsigned long long *>(out), len, slot);
        return hipGetLastError();
    }
#endif
    int v = c->enc_variant;
    if (v >= kQuadFirst) v = kDefaultEnc; // a quad variant asked for unaligned buffers: the default kernel handles any alignment
    // the LDS-transpose variant needs 16-byte aligned buffers on both sides
    if (variant_info(v).xpose && !(in_al && out_al)) v = kDefaultEnc;
    switch (v) {
#define X(id, U, B, NL, NS, XP, XC) \
    case id: return launch_encode_t<U, B, NL, NS, XP, XC>(c, seq, o, len, slot, XP ? true : in_al);
        BITNUC_VARIANTS(X)
#undef X
    default: return hipErrorInvalidValue;
    }
}

template <int UNROLL, int BLOCK, bool NTLD, bool NTST, bool XPOSE, bool XCD>
hipError_t launch_decode_t(bitnuc_ctx *c, const uint32_t *in32, uint8_t *out, unsigned long long n_bases, bool al) {
    const unsigned long long tile = (unsigned long long)BLOCK * UNROLL;
    const unsigned grid = grid_for(c, (n_bases >> 4) / tile + 1, BLOCK);
    if (al) decode_kernel<UNROLL, BLOCK, NTLD, NTST, true, XPOSE, XCD><<<grid, BLOCK, knobs(c).dyn_lds, c->stream>>>(in32, out, n_bases);
    else decode_kernel<UNROLL, BLOCK, NTLD, NTST, false, XPOSE, XCD><<<grid, BLOCK, knobs(c).dyn_lds, c->stream>>>(in32, out, n_bases);
    return hipGetLastError();
}

constexpr int kX2First = 47, kX2Last = 54; // decode variants of the evidence build's decode_x2_kernel
#ifdef BITNUC_SWEEP_VARIANTS
// decode variants 47..54: decode_x2_kernel (8-byte loads + LDS transpose) for the whole 2 KiB wave tiles, the default
// decode_kernel for what is left.  id - 47: bit 0 = nt loads, bit 1 = plain (not nt) stores, bit 2 = 2 words in flight per lane.
template <int UNROLL>
hipError_t launch_decode_x2_t(bitnuc_ctx *c, int mode, const unsigned long long *w, uint8_t *out, unsigned long long tiles) {
    constexpr int B = 256;
    const unsigned long long per = (unsigned long long)(B / 64) * UNROLL;
    const unsigned grid = (unsigned)((tiles + per - 1) / per);
    switch (mode & 3) {
    case 0: decode_x2_kernel<B, UNROLL, false, true><<<grid, B, 0, c->stream>>>(w, out, tiles); break;
    case 1: decode_x2_kernel<B, UNROLL, true, true><<<grid, B, 0, c->stream>>>(w, out, tiles); break;
    case 2: decode_x2_kernel<B, UNROLL, false, false><<<grid, B, 0, c->stream>>>(w, out, tiles); break;
    default: decode_x2_kernel<B, UNROLL, true, false><<<grid, B, 0, c->stream>>>(w, out, tiles); break;
    }
    return hipGetLastError();
}
#endif

hipError_t launch_decode(bitnuc_ctx *c, const uint64_t *ebuf, uint8_t *out, unsigned long long n_bases) {
    const bool in_al = aligned16(ebuf), out_al = aligned16(out);
#ifdef BITNUC_SWEEP_VARIANTS
    if (c->dec_variant >= kX2First && c->dec_variant <= kX2Last && out_al) {
        const unsigned long long tiles = n_bases >> 11; // whole 2 KiB (64-word) wave tiles
        if (tiles) {
            const int mode = c->dec_variant - kX2First;
            const unsigned long long *w = reinterpret_cast<const unsigned long long *>(ebuf);
            const hipError_t rc = (mode & 4) ? launch_decode_x2_t<2>(c, mode, w, out, tiles) : launch_decode_x2_t<1>(c, mode, w, out, tiles);
            if (rc != hipSuccess) return rc;
        }
        const unsigned long long done = tiles << 11;
        if (done == n_bases) return hipSuccess;
        return launch_decode_t<2, 256, false, true, false, false>(c, reinterpret_cast<const uint32_t *>(ebuf) + (done >> 4), out + done, n_bases - done, true);
    }
#endif
    const uint32_t *i = reinterpret_cast<const uint32_t *>(ebuf);
    int v = c->dec_variant;
    if (v >= kX2First) v = kDefaultDec; // x2 asked for an unaligned output: the default kernel handles any alignment
    if (variant_info(v).xpose && !in_al) v = kDefaultDec;
    switch (v) {
#define X(id, U, B, NL, NS, XP, XC) \
    case id: return launch_decode_t<U, B, NL, NS, XP, XC>(c, i, out, n_bases, out_al);
        BITNUC_VARIANTS(X)
#undef X
    default: return hipErrorInvalidValue;
    }
}
} // namespace

namespace bitnuc_rt {
bool codec_variant_built(int id) { return (kEvidenceBuild && id >= kQuadFirst && id <= kQuadLast) || variant_info(id).built; }
bool codec_decode_variant_ok(int id) { return (kEvidenceBuild && id >= kX2First && id <= kX2Last) || variant_info(id).built; }
int codec_num_variants() { return kNumVariants; }
int codec_ballot_variant() { return kEvidenceBuild ? kBallotVariant : -1; }
} // namespace bitnuc_rt

// ---- pipelined host-pointer path: host_pipe.h -------------------------------------------------------
namespace {

// encode / decode of a large pageable buffer through the engine of host_pipe.h.
// On INVALID_BASE the pipeline has still run every chunk and handed every chunk's words back: out[*n_words ..] is
// unspecified (include/bitnuc_hip.h), the words before the failing 32-base group are the reference's Vec contents.
int encode_pipelined(bitnuc_ctx *c, const uint8_t *seq, size_t len, uint64_t *out, size_t *n_words, bitnuc_err *err) {
    HostPipe *p;
    if (int st = pipe_get(c, &p, err)) return st;
    PipeAbort guard{c, p};
    struct Job {
        bitnuc_ctx *c; const uint8_t *seq; size_t len, chunk; uint64_t *out;
        size_t nchunks; int in_kind = kBufA, out_kind = kBufB, in_threads, out_threads;
        size_t bases(size_t ci) const { return len - ci * chunk < chunk ? len - ci * chunk : chunk; }
        const void *in_src(size_t ci) const { return seq + ci * chunk; }
        size_t in_bytes(size_t ci) const { return bases(ci); }
        void *out_dst(size_t ci) const { return out + ci * (chunk / 32); }
        size_t out_bytes(size_t ci) const { return words_for(bases(ci)) * 8; }
        int launch(size_t ci, const uint8_t *d_in, uint8_t *d_out, bitnuc_err *err) const {
            unsigned long long *slot;
            if (int st = take_slot(c, ci * chunk, &slot, err)) return st;
            HIPCHK(launch_encode(c, d_in, reinterpret_cast<uint64_t *>(d_out), bases(ci), slot));
            return BITNUC_OK;
        }
    } job{c, seq, len, p->chunk, out, (len + p->chunk - 1) / p->chunk};
    job.in_threads = p->enc_in;
    job.out_threads = p->enc_out;
    if (int st = pipe_run(c, p, job, err)) return st;
    bitnuc_err e;
    const int st = drain(c, &e); // one drain at the end: slots are examined in launch order = sequence order
    if (st == BITNUC_BACKEND_ERROR) { if (err) *err = e; return st; }
    guard.dismissed = true; // every stream has been waited for (the D2H events by the host, the kernels by the drain)
    if (st != BITNUC_OK) {
        if (err) *err = e;
        if (st == BITNUC_INVALID_BASE && n_words) *n_words = (size_t)(e.index / 32);
        return st;
    }
    if (n_words) *n_words = words_for(len);
    return BITNUC_OK;
}

int decode_pipelined(bitnuc_ctx *c, const uint64_t *ebuf, size_t n_bases, uint8_t *out, bitnuc_err *err) {
    HostPipe *p;
    if (int st = pipe_get(c, &p, err)) return st;
    PipeAbort guard{c, p};
    struct Job {
        bitnuc_ctx *c; const uint64_t *ebuf; size_t n_bases, chunk; uint8_t *out;
        size_t nchunks; int in_kind = kBufB, out_kind = kBufA, in_threads, out_threads;
        size_t bases(size_t ci) const { return n_bases - ci * chunk < chunk ? n_bases - ci * chunk : chunk; }
        const void *in_src(size_t ci) const { return ebuf + ci * (chunk / 32); }
        size_t in_bytes(size_t ci) const { return words_for(bases(ci)) * 8; }
        void *out_dst(size_t ci) const { return out + ci * chunk; }
        size_t out_bytes(size_t ci) const { return bases(ci); }
        int launch(size_t ci, const uint8_t *d_in, uint8_t *d_out, bitnuc_err *err) const {
            HIPCHK(launch_decode(c, reinterpret_cast<const uint64_t *>(d_in), d_out, bases(ci)));
            return BITNUC_OK;
        }
    } job{c, ebuf, n_bases, p->chunk, out, (n_bases + p->chunk - 1) / p->chunk};
    job.in_threads = p->dec_in;
    job.out_threads = p->dec_out;
    if (int st = pipe_run(c, p, job, err)) return st;
    HIPCHK(hipStreamSynchronize(c->stream));
    guard.dismissed = true;
    return BITNUC_OK;
}

} // namespace

namespace bitnuc_rt {

void pipe_destroy(HostPipe *p) { pipe_free(p); }

int encode_dev_at(bitnuc_ctx *c, const uint8_t *d_seq, size_t len, uint64_t *d_out, unsigned long long index_base, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (len == 0) return BITNUC_OK; // 0 words (the reference panics: packing/avx.rs:138)
    if (!d_seq || !d_out || (reinterpret_cast<uintptr_t>(d_out) & 7)) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    unsigned long long *slot;
    if (int st = take_slot(c, index_base, &slot, err)) return st;
    HIPCHK(launch_encode(c, d_seq, d_out, len, slot));
    return BITNUC_OK;
}

} // namespace bitnuc_rt

// =====================================================================================
// C ABI
// =====================================================================================
extern "C" {

// Diagnostic (bench.py's host_path block): how the pipelined host-pointer path of this context is configured.  Creates the pipe if
// this context has none yet.  out[0..n): cores_visible, cores_quota (0 = none), cores_usable, chunk_bases, depth, enc_in, enc_out,
// dec_in, dec_out (copy threads), heavy_cap, the GPU's NUMA node (-1 = unknown), CPUs of that node the workers are bound to (0 = free).
int bitnuc_host_pipe_info(bitnuc_ctx *c, double *out, int n, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (!out || n < 1) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    HostPipe *p;
    if (int st = pipe_get(c, &p, err)) return st;
    const double v[13] = {(double)p->cores_visible, (double)p->cores_quota, (double)p->cores_usable, (double)p->chunk, (double)kPipeDepth,
                          (double)p->enc_in, (double)p->enc_out, (double)p->dec_in, (double)p->dec_out, (double)p->heavy_cap,
                          (double)p->numa_node, (double)p->bound_cpus, (double)c->pipe_impl};
    for (int i = 0; i < n; ++i) out[i] = i < 13 ? v[i] : 0.0;
    return BITNUC_OK;
}

int bitnuc_encode_dev(bitnuc_ctx *c, const uint8_t *d_seq, size_t len, uint64_t *d_out, bitnuc_err *err) {
    return encode_dev_at(c, d_seq, len, d_out, 0, err);
}

int bitnuc_decode_dev(bitnuc_ctx *c, const uint64_t *d_ebuf, size_t n_words, size_t n_bases, uint8_t *d_out, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    // unpacking/mod.rs:40-45: missing words -> InvalidLength(n_bases)
    if (n_words < words_for(n_bases)) return fail(err, BITNUC_INVALID_LENGTH, n_bases);
    if (n_bases == 0) return BITNUC_OK; // unpacking/avx.rs:134-145: nothing appended
    if (!d_ebuf || !d_out || (reinterpret_cast<uintptr_t>(d_ebuf) & 7)) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    HIPCHK(launch_decode(c, d_ebuf, d_out, n_bases));
    return BITNUC_OK;
}
int bitnuc_nucgen_dev(bitnuc_ctx *c, uint8_t *d_out, size_t len, uint64_t seed, uint64_t first, int flags, bitnuc_err *err) {
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    if (len == 0) return BITNUC_OK;
    if (!d_out) return fail(err, BITNUC_UNSUPPORTED);
    DeviceGuard g(c->device);
    const unsigned grid = grid_for(c, ((len + 15) / 16 + kBlock - 1) / kBlock);
    nucgen_kernel<<<grid, kBlock, 0, c->stream>>>(d_out, len, seed, first, flags);
    HIPCHK(hipGetLastError());
    return BITNUC_OK;
}

int bitnuc_stream_probe_dev(bitnuc_ctx *c, int mode, const void *d_src, void *d_dst, size_t bytes, bitnuc_err *err) {
    // mode: bits 0-2 = 0 read / 1 copy / 2 fill; bit 3 = nt loads; bit 4 = nt stores; bit 5 = 2 (not 4) groups per lane
    clear_err(err);
    if (int st = check_ctx(c, err)) return st;
    DeviceGuard g(c->device);
    const unsigned long long n16 = bytes / 16;
    const bool ntl = (mode & 8) != 0, nts = (mode & 16) != 0, u2 = (mode & 32) != 0;
    const unsigned grid = grid_for(c, n16 / (kBlock * (u2 ? 2 : 4)) + 1);
    const u32x4 *src = static_cast<const u32x4 *>(d_src);
    u32x4 *dst = static_cast<u32x4 *>(d_dst);
#define PROBE(K, ...) K<<<grid, kBlock, 0, c->stream>>>(__VA_ARGS__)
    switch (mode & 7) {
    case 0:
        if (!d_src || !aligned16(d_src)) return fail(err, BITNUC_UNSUPPORTED);
        if (u2) { if (ntl) PROBE((probe_read_kernel<2, true>), src, n16, c->d_sink); else PROBE((probe_read_kernel<2, false>), src, n16, c->d_sink); }
        else { if (ntl) PROBE((probe_read_kernel<4, true>), src, n16, c->d_sink); else PROBE((probe_read_kernel<4, false>), src, n16, c->d_sink); }
        break;
    case 1:
        if (!d_src || !d_dst || !aligned16(d_src) || !aligned16(d_dst)) return fail(err, BITNUC_UNSUPPORTED);
        if (ntl && nts) PROBE((probe_copy_kernel<4, true, true>), src, dst, n16);
        else if (ntl) PROBE((probe_copy_kernel<4, true, false>), src, dst, n16);
        else if (nts) PROBE((probe_copy_kernel<4, false, true>), src, dst, n16);
        else PROBE((probe_copy_kernel<4, false, false>), src, dst, n16);
        break;
    case 2:
        if (!d_dst || !aligned16(d_dst)) return fail(err, BITNUC_UNSUPPORTED);
        if (nts) PROBE((probe_fill_kernel<4, true>), dst, n16); else PROBE((probe_fill_kernel<4, false>), dst, n16);
        break;
    case 3: { // encode_kernel's shape (variant 39: 2 rounds, 128 threads, nt loads, nt stores, XCD-contiguous tiles): `bytes` of ASCII-side input
        if (!d_src || !d_dst || !aligned16(d_src) || !aligned16(d_dst)) return fail(err, BITNUC_UNSUPPORTED);
        const unsigned g3 = grid_for(c, n16 / (128 * 2) + 1, 128);
        probe_enc_shape_kernel<2, 128, true, true, true><<<g3, 128, 0, c->stream>>>(src, static_cast<uint32_t *>(d_dst), n16);
        break;
    }
    case 4: { // decode_kernel's shape (variant 22: 2 rounds, 256 threads, plain loads, nt stores): `bytes` of ASCII-side output
        if (!d_src || !d_dst || !aligned16(d_src) || !aligned16(d_dst)) return fail(err, BITNUC_UNSUPPORTED);
        const unsigned g4 = grid_for(c, n16 / (256 * 2) + 1, 256);
        probe_dec_shape_kernel<2, 256, false, true><<<g4, 256, 0, c->stream>>>(static_cast<const uint32_t *>(d_src), dst, n16);
        break;
    }
#ifdef BITNUC_SWEEP_VARIANTS
    case 5: { // the every-window kernel's shape: `bytes` of ASCII-side input, 8 x as many bytes written.  bit 4: nt stores, bit 5: interleaved map, bits 6-7: rounds per trip 1 / 2 / 4
        if (!d_src || !d_dst || !aligned16(d_src) || !aligned16(d_dst)) return fail(err, BITNUC_UNSUPPORTED);
        const unsigned long long rounds = (bytes >> 10) & ~3ull;
        const int U = 1 << ((mode >> 6) & 3);
        const unsigned g5 = grid_for(c, (rounds + (unsigned long long)U * 4 - 1) / ((unsigned long long)U * 4));
#define WIN(NS, UU, MP) probe_win_shape_kernel<NS, UU, MP><<<g5, kBlock, 0, c->stream>>>(src, dst, rounds)
#define WIN_U(NS, MP) do { if (U == 1) WIN(NS, 1, MP); else if (U == 2) WIN(NS, 2, MP); else WIN(NS, 4, MP); } while (0)
        if (mode & 32) { if (nts) WIN_U(true, 1); else WIN_U(false, 1); }
        else { if (nts) WIN_U(true, 0); else WIN_U(false, 0); }
#undef WIN_U
#undef WIN
        break;
    }
#endif
    default:
        return fail(err, BITNUC_UNSUPPORTED);
    }
#undef PROBE
    HIPCHK(hipGetLastError());
    return BITNUC_OK;
}

// ---- host-pointer entry points (synchronous; size dispatch: runtime.h on_host) -------------------
int bitnuc_encode(bitnuc_ctx *c, const uint8_t *seq, size_t len, uint64_t *out, size_t *n_words, bitnuc_err *err) {
    clear_err(err);
    if (n_words) *n_words = 0;
    if (len == 0) return BITNUC_OK; // 0 words (the reference panics there: packing/avx.rs:138)
    if (!seq || !out) return fail(err, BITNUC_UNSUPPORTED);
    if (on_host(c, len)) { // host_word.h: same words, same first-invalid-byte rule, no launch
        const long long bad = bitnuc_host::encode_small(seq, len, out);
        if (bad >= 0) {
            if (err) { memset(err, 0, sizeof *err); err->status = BITNUC_INVALID_BASE; err->byte = seq[bad]; err->index = (uint64_t)bad; }
            if (n_words) *n_words = (size_t)bad / 32;
            return BITNUC_INVALID_BASE;
        }
        if (n_words) *n_words = words_for(len);
        return BITNUC_OK;
    }
    if (int st = check_ctx(c, err)) return st;
    DeviceGuard g(c->device);
    if (int st = flush_pending(c, err)) return st;
    if (c->host_pipeline && len >= kPipeMin) return encode_pipelined(c, seq, len, out, n_words, err);
    const size_t chunk = len < kHostChunk ? len : kHostChunk;
    if (int st = ensure_scratch(c, 0, chunk + 16, err)) return st;
    if (int st = ensure_scratch(c, 1, words_for(chunk) * 8 + 16, err)) return st;
    for (size_t off = 0; off < len; off += chunk) {
        const size_t n = len - off < chunk ? len - off : chunk;
        const size_t nw = words_for(n);
        HIPCHK(hipMemcpyAsync(c->scratch[0], seq + off, n, hipMemcpyHostToDevice, c->stream));
        unsigned long long *slot;
        if (int st = take_slot(c, off, &slot, err)) return st;
        HIPCHK(launch_encode(c, c->scratch[0], reinterpret_cast<uint64_t *>(c->scratch[1]), n, slot));
        HIPCHK(hipMemcpyAsync(out + off / 32, c->scratch[1], nw * 8, hipMemcpyDeviceToHost, c->stream));
        // the call is synchronous and stops at the first failing chunk (the words before it are the caller's)
        bitnuc_err e;
        int st = drain(c, &e);
        if (st != BITNUC_OK) {
            if (err) *err = e;
            if (st == BITNUC_INVALID_BASE && n_words) *n_words = (size_t)(e.index / 32);
            return st;
        }
    }
    if (n_words) *n_words = words_for(len);
    return BITNUC_OK;
}

int bitnuc_decode(bitnuc_ctx *c, const uint64_t *ebuf, size_t n_words, size_t n_bases, uint8_t *out, bitnuc_err *err) {
    clear_err(err);
    if (n_words < words_for(n_bases)) return fail(err, BITNUC_INVALID_LENGTH, n_bases);
    if (n_bases == 0) return BITNUC_OK;
    if (!ebuf || !out) return fail(err, BITNUC_UNSUPPORTED);
    if (on_host(c, n_bases, true)) {
        bitnuc_host::decode_small(ebuf, n_bases, out);
        return BITNUC_OK;
    }
    if (int st = check_ctx(c, err)) return st;
    DeviceGuard g(c->device);
    // like every host-pointer call: an InvalidBase latched by earlier asynchronous launches stays for the next bitnuc_ctx_sync
    // (decode itself latches nothing, but the pipeline's abort path drains the ring and would drop it)
    if (int st = flush_pending(c, err)) return st;
    if (c->host_pipeline && n_bases >= kPipeMin) return decode_pipelined(c, ebuf, n_bases, out, err);
    const size_t chunk = n_bases < kHostChunk ? n_bases : kHostChunk;
    if (int st = ensure_scratch(c, 0, chunk + 16, err)) return st;
    if (int st = ensure_scratch(c, 1, words_for(chunk) * 8 + 16, err)) return st;
    for (size_t off = 0; off < n_bases; off += chunk) {
        const size_t n = n_bases - off < chunk ? n_bases - off : chunk;
        const size_t nw = words_for(n);
        HIPCHK(hipMemcpyAsync(c->scratch[1], ebuf + off / 32, nw * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(launch_decode(c, reinterpret_cast<const uint64_t *>(c->scratch[1]), c->scratch[0], n));
        HIPCHK(hipMemcpyAsync(out + off, c->scratch[0], n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return BITNUC_OK;
}
// ---- single-word API: host code; batches of one on the device when forced ----------------
int bitnuc_as_2bit(bitnuc_ctx *c, const uint8_t *seq, size_t len, uint64_t *out, bitnuc_err *err) {
    clear_err(err);
    if (len > 32) return fail(err, BITNUC_SEQUENCE_TOO_LONG, len); // packing/naive.rs:5-7: before any base is looked at
    if (!out || (len && !seq)) return fail(err, BITNUC_UNSUPPORTED);
    if (c && c->force_gpu) return bitnuc_as_2bit_batch(c, seq, len, len ? len : 1, 1, out, err);
    uint64_t w = 0;
    const int bad = bitnuc_host::pack_word(seq, len, &w);
    if (bad >= 0) {
        if (err) { memset(err, 0, sizeof *err); err->status = BITNUC_INVALID_BASE; err->byte = seq[bad]; err->index = (uint64_t)bad; }
        return BITNUC_INVALID_BASE;
    }
    *out = w;
    return BITNUC_OK;
}

int bitnuc_from_2bit(bitnuc_ctx *c, uint64_t packed, size_t n, uint8_t *out, bitnuc_err *err) {
    clear_err(err);
    if (n > 32) return fail(err, BITNUC_INVALID_LENGTH, n); // unpacking/naive.rs:8-10
    if (n == 0) return BITNUC_OK;
    if (!out) return fail(err, BITNUC_UNSUPPORTED);
    if (c && c->force_gpu) return bitnuc_decode(c, &packed, 1, n, out, err);
    bitnuc_host::unpack_word(packed, n, out);
    return BITNUC_OK;
}

int bitnuc_hdist_scalar(bitnuc_ctx *c, uint64_t u, uint64_t v, size_t len, uint32_t *out, bitnuc_err *err) {
    clear_err(err);
    if (len > 32) return fail(err, BITNUC_INVALID_LENGTH, len); // hamming/scalar.rs:13-15
    if (!out) return fail(err, BITNUC_UNSUPPORTED);
    if (c && c->force_gpu) return bitnuc_hdist(c, &u, 1, &v, 1, len, out, err);
    *out = bitnuc_host::hdist_word(u, v, len);
    return BITNUC_OK;
}

// Diagnostic (tools/host_path.py): GB/s of the staging pool's parallel memcpy of `bytes` with `threads` threads;
// mode 0: pageable -> pageable, 1: pageable -> pinned (hipHostMalloc), 2: pinned -> pageable.  < 0 on failure.
double bitnuc_selftime_host_copy(size_t bytes, int threads, int mode) {
    if (bytes < 4096 || threads < 1 || threads > 64 || mode < 0 || mode > 2) return -1.0;
    uint8_t *page = static_cast<uint8_t *>(malloc(bytes)), *other = nullptr;
    if (!page) return -1.0;
    memset(page, 65, bytes);
    if (mode == 0) { other = static_cast<uint8_t *>(malloc(bytes)); if (other) memset(other, 1, bytes); }
    else if (hipHostMalloc(reinterpret_cast<void **>(&other), bytes, hipHostMallocDefault) != hipSuccess) other = nullptr;
    if (!other) { free(page); return -1.0; }
    if (mode != 0) memset(other, 1, bytes);
    double best = 0.0;
    {
        CopyPool pool(threads);
        for (int rep = 0; rep < 4; ++rep) {
            struct timespec t0, t1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            if (mode == 2) pool.copy(page, other, bytes); else pool.copy(other, page, bytes);
            clock_gettime(CLOCK_MONOTONIC, &t1);
            const double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
            if (rep > 0 && (double)bytes / sec / 1e9 > best) best = (double)bytes / sec / 1e9;
        }
    }
    if (mode == 0) free(other); else (void)hipHostFree(other);
    free(page);
    return best;
}

// Diagnostic (bench.py's small_call_latency block): mean ns per call of the HOST path over `iters` calls on the
// reference's bench input (cyclic "ACGT", benches/simd_comparison.rs:4-7), timed here so that no binding overhead is in it.
// op: 0 as_2bit, 1 from_2bit, 2 encode, 3 decode, 4 hdist_scalar.  Returns < 0 on a bad argument.
double bitnuc_selftime_small(int op, size_t n, size_t iters) {
    if (iters == 0 || n == 0 || op < 0 || op > 4 || ((op == 0 || op == 1 || op == 4) && n > 32)) return -1.0;
    // only sizes the host path takes with ctx == NULL (everything here is called without a context)
    if (n >= kDefaultHostCutoffDecode || (op == 2 && n >= kDefaultHostCutoff)) return -1.0;
    std::vector<uint8_t> seq(n), back(n);
    for (size_t i = 0; i < n; ++i) seq[i] = "ACGT"[i & 3];
    std::vector<uint64_t> words(words_for(n) + 1);
    size_t nw = 0;
    bitnuc_err e;
    if (bitnuc_encode(nullptr, seq.data(), n, words.data(), &nw, &e) != BITNUC_OK) return -1.0;
    volatile uint64_t sink = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (size_t it = 0; it < iters; ++it) {
        uint64_t w = 0;
        uint32_t d = 0;
        seq[0] = "AC"[it & 1]; // the input changes between calls: the compiler cannot hoist the work
        switch (op) {
        case 0: (void)bitnuc_as_2bit(nullptr, seq.data(), n, &w, &e); sink += w; break;
        case 1: (void)bitnuc_from_2bit(nullptr, words[0] ^ it, n, back.data(), &e); sink += back[0]; break;
        case 2: (void)bitnuc_encode(nullptr, seq.data(), n, words.data(), &nw, &e); sink += words[0]; break;
        case 3: words[0] ^= it & 3; (void)bitnuc_decode(nullptr, words.data(), nw, n, back.data(), &e); sink += back[0]; break;
        case 4: (void)bitnuc_hdist_scalar(nullptr, words[0] ^ it, words[0], n, &d, &e); sink += d; break;
        default: return -1.0;
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    (void)sink;
    return ((double)(t1.tv_sec - t0.tv_sec) * 1e9 + (double)(t1.tv_nsec - t0.tv_nsec)) / (double)iters;
}

} // extern "C"
