// runtime.h -- internal interface between the translation units of libbitnuc_hip.so (not installed, not part of the C ABI):
//   runtime.hip   the context: device + stream, error slots and their lifetime, scratch, knobs       (this header's functions)
//   codec.hip     bulk encode / decode, the single-word API, the pipelined host-pointer path, probes  (codec_device.h)
//   kmer.hip      k-mer batches, sliding scan (matrix cores), bulk hdist                             (kmer_device.h, scan_mfma_device.h)
//   batch.hip     ragged / planned / fixed-length batches of reads                                   (batch_device.h)
//   analysis.hip  base counts, many-pair hdist, split_packed                                         (analysis_device.h)
//   comm.hip      RCCL all-gather of the packed words, xGMI link probe
// Each kernel header is included by exactly one of them; what they share on the device is device_prims.h.
// csrc/evidence/*.h hold the kernels that lost their A/B: the kernel headers include them only under -DBITNUC_SWEEP_VARIANTS
// (libbitnuc_hip_sweep.so); the product library never sees them.
#pragma once
#include "../../include/bitnuc_hip.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <vector>

struct HostPipe; // codec.hip

namespace bitnuc_rt {

constexpr int kSlotBlock0 = 4096;        // error slots of a fresh context (launches between two syncs before the ring grows)
constexpr int kSlotBlockMax = 1 << 20;   // the largest block the ring grows to (see take_slot)
constexpr int kCapturedSlots = 1024;     // persistent slots of launches recorded into a hipGraph (a context's lifetime total)
constexpr size_t kHostChunk = (size_t)128 << 20; // bases per staged chunk on the simple host-pointer path
// Size dispatch of the host-pointer bulk calls: the measured crossover between the library's host SWAR code (one
// thread, 9-10 Gbases/s) and the GPU path (stage in, launch, stage out, one wait: 33-40 us + PCIe) on the GPU box, tools/host_cutoff.py,
// profiles/r02_host_cutoff.txt: encode 1 Mi bases (101 vs 104 us), decode 512 Ki bases (62 vs 63 us).
constexpr size_t kDefaultHostCutoff = (size_t)1 << 20;       // encode, hdist
constexpr size_t kDefaultHostCutoffDecode = (size_t)1 << 19; // decode
constexpr int kDefaultEnc = 39, kDefaultDec = 22;            // codec.hip: the variant tables

// Selectors of ALTERNATIVE FORMULATIONS -- the ones that lost their A/B (profiles/README.md) and timing-only ablations.  The product
// context does not hold them: knobs(c) is this struct's defaults, a compile-time constant, and bitnuc_ctx_set_variant() answers -2 to
// anything but the shipped value.  Only the evidence build (-DBITNUC_SWEEP_VARIANTS, libbitnuc_hip_sweep.so: tools/ and the variant
// tests) keeps a copy per context that can be changed.  The default of every field IS what ships.
struct SweepKnobs {
    int dyn_lds = 0;             // bytes of unused dynamic LDS per workgroup of the bulk codec launches (an occupancy limiter, tools/ab_occupancy.py)
    int batch_dense = 1;         // stride == k batches use kmer_dense_kernel (0: the general strided kernel)
    int batch_slide = 1;         // stride 1 (every window of a sequence), 2, 4, 8, 16 batches use the sliding kernels (0: the general strided kernel)
    int slide_impl = 1;          // stride-1 windows: 1 = line-aligned rounds of 1024 windows (kmer_slide2_kernel), 0 = rounds of 992 (kmer_slide_kernel)
    int slide_rounds = 1;        // kmer_slide_kernel: consecutive rounds per wave trip (1, 2, 4, 8)
    int slide2_rounds = 4;       // kmer_slide2_kernel: consecutive 1 KiB rounds per wave trip (1, 2, 4)
    int dense_policy = 3, scan_policy = 3; // bit0: nt loads, bit1: nt stores
    int kmer_block = 256;        // threads per workgroup of the dense-batch and scan kernels: 64, 128 or 256
    int dense_unroll = 1;        // items (64 k-mers = 2 dwordx4 per lane) in flight per wave: 1, 2 or 4
    int scan_unroll = 4;         // rounds (1 KiB loads) in flight per wave: 1, 2 or 4
    int scan_impl = 8;           // the one-hot contraction on the matrix cores: 8 = in the fused count's tiling (kmer_scan_seg_mfma_kernel: four MFMAs per 1024 windows + two v_permlane32_swap, ships:
                                 // profiles/r05_ab_scan_seg.txt), 7 = the natural-layout tiling (kmer_scan_mfma_kernel: six MFMAs, round 5's first form, profiles/r05_ab_scan_mfma*.txt);
                                 // the bit-plane forms (v_alignbit + v_bcnt per window, VALU-issue bound): 1 = line-aligned rounds of 1024 windows, two-LUT planes + scalar halo
                                 // (kmer_scan2_kernel GEN 1: shipped in round 4), 6 = the same with rounds 2-3's plane build (GEN 0), 0 = rounds of 992 windows (kmer_scan_kernel),
                                 // 2 / 3 / 4 / 5 = kmer_scan3_kernel (a wave owns 12 / 20 / 16 / 32 consecutive rounds: slower, profiles/r04_ab_scan3.txt)
    int scan_mfma_unroll = 4;    // kmer_scan_mfma_kernel: consecutive 1 KiB rounds per wave trip: 2, 3 or 4 (profiles/r05_ab_scan_trip_length.txt)
    int scan_mfma_grid = 4;      // ... bounded-grid form (scan_mfma_persist 1): workgroups per CU (profiles/r05_ab_scan_grid*.txt: flatter from idle, 2.5-5 % slower settled)
    int scan_mfma_policy = 3;    // ... bit 0: nt loads (stores are nt)
    int scan_mfma_shift = 4;     // ... the shifted operands: 0 = two more global loads, 1 = the bytes through a wave-private LDS strip, 2 = DPP + scalar halo,
                                 //     3 = the one-hot operands through the strip, 4 = ... with the lane's own operands kept in registers (ships)
    int scan_mfma_block = 64;        // ... kmer_scan_seg_mfma_kernel: threads per workgroup: 64 (one wave: ships), 128, 256 (trips of four rounds only)
    int scan_mfma_ch3 = 0;           // ... the shipped scan's tiling with three channels per base instead of four (kmer_scan_seg3_mfma_kernel: three MFMAs per 1024 windows), evidence build
    int scan_mfma_match = 0;         // ... the query's side of the product: 0 = 1.0 on the channels that DIFFER from the query's base (three of four), 1 = -1.0 on the one that EQUALS it, counted down from k (a third of the non-zero entries)
    int scan_mfma_count_emit = 2;    // ... its own tiling's results -> count: 0 = v_cmp + s_bcnt1 per register, 1 = threshold fields inside the product (v_or3 + v_bitop3 + v_bcnt per four windows), 2 = 1 + the next trip loaded into the same registers (ships)
    int scan_mfma_count_form = 2;    // ... the fused count: 2 = segments of 32 windows with three channels per base (kmer_count3_mfma_kernel: 3 MFMAs per 1024 windows, ships), 1 = four channels (kmer_count_mfma_kernel: 4 MFMAs), 0 = the scan's natural-layout tiling (6)
    int scan_mfma_count_rounds = 4;  // kmer_count3_mfma_kernel / kmer_count_mfma_kernel: rounds per trip (2, 3, 4); the four-channel form shipped with 3 (six waves share a SIMD)
    int scan_mfma_count_grid = 12;   // ... workgroups per CU (the four-channel form shipped with 18: three generations of the six resident ones)
    int scan_mfma_count_persist = 1; // ... the fused count: 1 = a resident grid (ships), 0 = one trip per wave (two atomics per workgroup at the accumulator and the ticket)
    int scan_mfma_persist = 0;   // ... 1 = a resident grid walks the trips with register prefetch, 0 = one trip per wave
    int scan_mfma_pack = 1;      // ... f32 -> u8: 0 = v_cvt_pk_u8_f32, 1 = 2^23 bias + row scales (copied), 2 = ... (bias by a seventh instruction)
    int hdist_tiled = 0;         // bulk hdist: 1 = grid-stride at tile granularity (16 KiB of each operand per workgroup trip), 0 = at thread granularity
    int hdist_words_impl = 1;    // many-pair / one-query hdist: 1 = coalesced loads + bpermute for whole 256-word tiles, 0 = four contiguous words per lane
    int fixed_stream = 1;        // encode_fixed (back-to-back reads): 1 = cut the tile's 2-bit stream
    int fixed_dec_strip = 2;     // decode_fixed (back-to-back reads): 0 = byte scatter, 1 = bit strip with per-lane 64-bit positions, 2 = the plan decode's tile body with arithmetic lookups
    int owner_est = 3;           // block_owner_kernel's first guess: 0 = 128-bit division, 1 = double, 2 = exact 0.64 fixed-point multiply-high, 3 = 2 or 0 by average sequence length
    int batch_tables_impl = 1;   // table-driven ragged batches: 1 = one asynchronous pass emits the layout plan into context scratch, then the plan kernels; 0 = tile records + O(1) lookup kernels
    int batch_host_plan = 1;     // host-pointer ragged-batch calls build a layout plan (bitnuc_batch_plan) and use the plan kernels (0: the table-driven form)
    int plan_dec_lines = 0;      // plan decode: 0 = word tiles with shared edge lines (decode_batch_plan_kernel: ships), 1 = line-owning, 2 = chunk-owning tiles (decode_batch_plan_lines_kernel: slower, profiles/r04_ab_plan_lines.txt)
    int plan_tiles = 1;          // decode_batch_plan_kernel: consecutive tiles per wave trip (1, 2 or 4)
    int plan_store = 2;          // decode_batch_plan_kernel's whole-chunk store policy: 0 nt, 1 plain, 2 plain on the shared edge lines + nt elsewhere
    int plan_enc_block = 256;    // threads per workgroup of the plan encode (64, 128, 256)
    int plan_enc_tiles = 1;      // encode_batch_plan_kernel: consecutive tiles per wave trip (1, 2 or 4)
    int plan_enc_abl = 0;        // timing-only ablations of the plan encode's loads (see plan_enc_issue)
    int batch_abl = 0;           // timing-only ablation mask of the second table-driven formulation (tools/ab_batch_ablate.py)
};

// One block of per-launch error slots: device words (all-ones = no error) + a pinned mirror + the index base of each launch.
struct SlotBlock {
    unsigned long long *d = nullptr, *h = nullptr;
    int cap = 0, used = 0;
    std::vector<unsigned long long> base; // added to the slot's index (host path chunk offset)
};

} // namespace bitnuc_rt

struct bitnuc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cu = 256;
    // ---- data-error slots (runtime.hip: take_slot / drain) ----
    std::vector<bitnuc_rt::SlotBlock> slots;     // ordinary launches: blocks in launch order, the last one is being filled
    unsigned long long *d_cap = nullptr, *h_cap = nullptr; // launches recorded into a hipGraph: persistent, examined and re-armed by every drain
    unsigned long long cap_base[bitnuc_rt::kCapturedSlots];
    int n_cap = 0;
    std::vector<bitnuc_err> deferred; // data errors found by implicit drains (host-pointer calls start from an empty ring), oldest first: one per bitnuc_ctx_sync
    // ---- scratch ----
    uint8_t *scratch[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t scratch_cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool scratch_in_graph[8] = {false, false, false, false, false, false, false, false}; // handed to a launch recorded into a hipGraph: never freed before the context
    std::vector<uint8_t *> retired_scratch; // ... outgrown since: alive until bitnuc_ctx_destroy (a replay still writes through them)
    uint32_t *d_sink = nullptr;
    unsigned long long *d_acc = nullptr; // accumulators of the single-launch reductions, zero between launches: [0..2] base_counts C,G,T; [4] hdist (u32); [5] scan count
    unsigned *d_tickets = nullptr;       // [0] base_counts, [1] hdist, [2] scan count: arrival counters, zero between launches
    unsigned reduce_blocks = 512;        // base_counts: resident grid (2 workgroups per CU)
    unsigned hdist_blocks = 256;         // bulk hdist: resident grid (1 workgroup per CU: 8 loads in flight per thread; profiles/r03_ab_hdist_grid.txt)
    // ---- knobs (bitnuc_ctx_set_variant): what an integrator may turn ----
    int enc_variant = bitnuc_rt::kDefaultEnc, dec_variant = bitnuc_rt::kDefaultDec; // the bulk codec's shipped variants (codec.hip)
    int grid_mult = 0;                   // see grid_for()
#ifdef BITNUC_SWEEP_VARIANTS
    bitnuc_rt::SweepKnobs sweep;         // evidence build only: selectors of the formulations that lost their A/B (see SweepKnobs)
#endif
    bitnuc_batch_plan *host_plan = nullptr; // the layout plan the host-pointer ragged-batch calls build and keep
    int force_gpu = 0;                     // 1: single-word and below-cutoff calls launch kernels too (GPU parity tests, BITNUC_FORCE_GPU=1)
    size_t host_cutoff = bitnuc_rt::kDefaultHostCutoff; // bulk host-pointer encode / hdist below this many bases run on the host (host_word.h)
    size_t host_cutoff_decode = bitnuc_rt::kDefaultHostCutoffDecode; // ... decode
    int host_pipeline = 1;                 // large host-pointer encode / decode: pinned buffers + overlapped H2D / kernel / D2H
    int pipe_impl = 1;                     // host-pointer pipeline: 1 = direct engine (pageable copies from the calling thread + one mover thread: ships -- faster on four of four
                                           // boxes, profiles/r03_ab_pipe_impl.txt), 0 = staged engine (own pinned buffers + copy threads); BITNUC_PIPE_IMPL=direct|staged (csrc/host_pipe.h)
    HostPipe *pipe = nullptr;              // created on the first large host-pointer call (codec.hip)
};

namespace bitnuc_rt {

// ---- error values ---------------------------------------------------------------------------------------------------
inline void clear_err(bitnuc_err *e) {
    if (e) memset(e, 0, sizeof *e);
}
inline int fail(bitnuc_err *e, int status, uint64_t value = 0) {
    if (e) { memset(e, 0, sizeof *e); e->status = status; e->value = value; }
    return status;
}
inline int fail_hip(bitnuc_err *e, hipError_t rc) {
    if (e) { memset(e, 0, sizeof *e); e->status = BITNUC_BACKEND_ERROR; e->backend_code = (int32_t)rc; }
    return BITNUC_BACKEND_ERROR;
}
#define HIPCHK(expr)                                                   \
    do {                                                               \
        hipError_t rc__ = (expr);                                      \
        if (rc__ != hipSuccess) return ::bitnuc_rt::fail_hip(err, rc__); \
    } while (0)

struct DeviceGuard { // hipSetDevice is per-thread state: every entry point selects the context's device and restores the caller's
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
};

inline int check_ctx(bitnuc_ctx *c, bitnuc_err *err) {
    if (!c) return fail(err, BITNUC_UNSUPPORTED);
    return BITNUC_OK;
}

// ---- context services (runtime.hip) ---------------------------------------------------------------------------------
int ensure_scratch(bitnuc_ctx *c, int which, size_t bytes, bitnuc_err *err);
// Reserve the error slot of the next launch.  `base` is added to the byte index the kernel latches.  While the context's
// stream is being captured into a hipGraph the launch gets a PERSISTENT slot (it will run again at every replay); otherwise
// the next slot of the ring, which grows when it is full (no stream synchronisation inside an asynchronous call).
int take_slot(bitnuc_ctx *c, unsigned long long base, unsigned long long **slot, bitnuc_err *err);
// the index base of the most recently taken slot (host paths that learn their chunk offset after the launch call)
void set_last_slot_base(bitnuc_ctx *c, unsigned long long base);
// Wait for the stream, report the first latched data error (ordinary launches in launch order, then captured launches in
// capture order), re-arm what fired, empty the ring.
int drain(bitnuc_ctx *c, bitnuc_err *err);
// Host-pointer (synchronous) calls start from an empty ring so that the error they return is their own; an InvalidBase
// latched by earlier asynchronous launches is kept for the next bitnuc_ctx_sync().
int flush_pending(bitnuc_ctx *c, bitnuc_err *err);
bool slots_outstanding(const bitnuc_ctx *c);

// grid_mult = resident 256-thread workgroups per CU for grid-stride launches (scaled for other block sizes); 0 = one tile per
// workgroup, the hardware dispatcher walks the tiles (fastest for the streaming codec: no tail imbalance -- profiles/).
inline unsigned grid_for(const bitnuc_ctx *c, unsigned long long tiles, int block = 256) {
    if (tiles == 0) return 1;
    if (c->grid_mult <= 0) return (unsigned)(tiles < 0x7FFFFFFFull ? tiles : 0x7FFFFFFFull);
    unsigned long long cap = (unsigned long long)c->num_cu * c->grid_mult * 256 / block;
    if (cap == 0) cap = 1;
    return (unsigned)(tiles < cap ? tiles : cap);
}

inline size_t words_for(size_t n_bases) { return n_bases / 32 + (n_bases % 32 != 0); } // ceil(n/32) without overflow
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// true when a bulk host-pointer call of n bases belongs on the host: below the cutoff and not forced to the GPU.
// A NULL context is accepted for such calls (the reference's functions need no context either).
inline bool on_host(const bitnuc_ctx *c, size_t n, bool decode = false) {
    if (c) return !c->force_gpu && n < (decode ? c->host_cutoff_decode : c->host_cutoff);
    return n < (decode ? kDefaultHostCutoffDecode : kDefaultHostCutoff);
}

// Alternative formulations that lost their A/B (profiles/) stay in the source as evidence, but only the evidence build
// (-DBITNUC_SWEEP_VARIANTS, libbitnuc_hip_sweep.so: tools/ and the variant tests) instantiates them; in the product a
// set_variant() to anything but the shipped value returns -2 and changes nothing.
#ifdef BITNUC_SWEEP_VARIANTS
constexpr bool kEvidenceBuild = true;
inline const SweepKnobs &knobs(const bitnuc_ctx *c) { return c->sweep; }
#else
constexpr bool kEvidenceBuild = false;
constexpr SweepKnobs kShipped{};
inline constexpr const SweepKnobs &knobs(const bitnuc_ctx *) { return kShipped; } // a constant: the launch code's branches on it fold away
#endif

// ---- cross-unit entry points that are not part of the C ABI ------------------------------------------------------------
// codec.hip
void pipe_destroy(HostPipe *p);
bool codec_variant_built(int id);            // encode / decode variant ids this build holds
bool codec_decode_variant_ok(int id);        // ... plus the evidence build's decode_x2 ids
int codec_num_variants();
int codec_ballot_variant();                  // -1 when this build does not hold it
// enqueue an encode of `len` bases whose latched byte index is reported as index_base + (offset inside d_seq)
int encode_dev_at(bitnuc_ctx *c, const uint8_t *d_seq, size_t len, uint64_t *d_out, unsigned long long index_base, bitnuc_err *err);

} // namespace bitnuc_rt
