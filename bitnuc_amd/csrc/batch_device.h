// batch_device.h -- ragged batches of independent sequences (reads, contigs) in one launch.
//
// The reference's user loops `encode(seq_i, &mut ebuf_i)` / `decode(&ebuf_i, len_i, &mut dbuf)`
// over many short sequences (src/utils/mod.rs:22-25,60-62; each call pads its own last word,
// packing/avx.rs:147-148).  Here the sequences sit back to back in one buffer with an
// offsets table, and their words back to back with a word-offsets table
// (word_offsets[i] = sum_{j<i} ceil(len_j/32)); sequence boundaries fall anywhere, so:
//   * one lane owns one output WORD; a WAVE owns 64 consecutive words, whose bytes are one
//     contiguous span of at most 2 KiB because sequences and their words are contiguous;
//     waves never wait for each other (no workgroup barrier anywhere in these kernels);
//   * the span moves between HBM and LDS with coalesced 16-byte accesses; lanes touch their
//     (unaligned, 1..32-byte) pieces in LDS only: encode funnels them out of the staged bytes
//     (or cuts the tile's 2-bit stream), decode ORs its words into a bit strip whose dwords are
//     the output's aligned 16-byte chunks;
//   * word -> sequence lookup: ONE PAD BYTE PER WORD and one base offset per 64-word tile (the layout plan: bitnuc_batch_plan, or
//     context scratch re-emitted by plan_emit_kernel for the table-driven entry points); a wave scan of the pad bytes gives every
//     lane its first base.  No search anywhere.
//
// WHAT SHIPS (the product library instantiates only these):
//   word_offsets_block_sums / _scan_sums / _finish      bitnuc_batch_word_offsets_dev (three-launch exclusive scan)
//   plan_emit_kernel                                     the layout plan from the offsets table(s), one pass
//   encode_batch_plan_kernel<1>, decode_batch_plan_kernel<2, 1>     ragged batches (plan and table-driven entry points)
//   encode_fixed_kernel, decode_fixed_tile_kernel<2>     fixed-length reads (the two lookups are arithmetic)
// EVIDENCE BUILD ONLY (-DBITNUC_SWEEP_VARIANTS; csrc/evidence/batch_evidence.h and the other instantiations of the templates below; formulations that lost their A/B, kept so that a result can be re-measured --
// profiles/NARRATIVE_r01_r03.md 3.5, profiles/README.md): block_owner_kernel + encode_batch2_kernel / decode_batch2_kernel
// (round 2: tile records by a search pre-kernel, O(1) pad-scatter lookup), decode_batch_kernel and decode_fixed_kernel /
// decode_fixed_strip_kernel (round 1-2 decode bodies), the U = 2 / 4 and ABL / POLICY instantiations of the plan kernels,
// decode_batch_plan_lines_kernel (round 4: line- / chunk-owning tiles).
#pragma once
#include "device_prims.h"

namespace bitnuc_dev {

constexpr int kBatchTile = 64;                      // words per WAVE: every wave works alone on its own tile, so the
                                                    // kernels have no workgroup barrier at all
constexpr int kBatchStage = kBatchTile * 32 + 128;  // 2 KiB span + alignment / read-ahead slack, per wave
constexpr int kBatchWaves = kBlock / 64;            // waves (= tiles in flight) per workgroup

// Cache policy of the PACKED-WORD stores of every read-batch encode (0.25 B per base): streaming (nt).  Allocating stores --
// what the bulk encode's variant 14 uses, where they won the PAIR sweep of round 1 -- make these kernels 3-13 % slower in
// encode-only bursts (profiles/r02_ab_encode_word_store_policy.txt).
constexpr bool kEncWordsNT = true;
template <class T> __device__ __forceinline__ void store_packed(T v, T *p) {
    if constexpr (kEncWordsNT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// The batch kernels load whole 16-byte ALIGNED chunks, so the chunk that holds the batch's first base may begin with up to 15
// bytes that are not the batch's (a FASTA header's newline, another buffer).  They are never validated -- but enc4 compacts a
// dword's four codes with a multiply-add, and a byte that is not a base leaves more than two bits there: a foreign byte in
// the SAME dword as the batch's first bases would spill into their codes (found in round 3 by the first test that handed the
// _dev entry points a batch starting mid-dword behind 'N' bytes; the host-pointer forms stage the batch at an aligned address and
// never saw it).  The first chunk of the first tile therefore has its leading foreign bytes replaced by 'A' (code 0, valid).
__device__ __forceinline__ u32x4 mask_lead_bytes(u32x4 v, unsigned lead /* 0..15 bytes to replace */) {
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int nbytes = (int)lead - 4 * i;
        const uint32_t m = nbytes >= 4 ? ~0u : (nbytes <= 0 ? 0u : (1u << (8 * nbytes)) - 1u);
        w[i] = (w[i] & ~m) | (0x41414141u & m);
    }
    return u32x4{w[0], w[1], w[2], w[3]};
}

// index of the sequence that owns word w: upper_bound(word_offsets[0..count], w) - 1
__device__ __forceinline__ unsigned long long owner_of_word(const unsigned long long *__restrict__ wo,
                                                            unsigned long long count, unsigned long long w) {
    unsigned long long lo = 0, hi = count + 1;
    while (lo < hi) {
        const unsigned long long mid = (lo + hi) >> 1;
        if (wo[mid] <= w) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}

// Write-out of one 16-byte stage chunk to global address g, restricted to [lo, hi): whole
// chunks are one dwordx4 store; a chunk cut by the span's edge (its other bytes belong to the
// neighbouring tile) is written as <= 3 bytes, <= 3 aligned dwords, <= 3 bytes -- predicated
// stores, no loop (the edge lanes would otherwise hold their whole wave for a 16-trip loop).
__device__ __forceinline__ void store_stage_chunk(const uint8_t *chunk, uintptr_t g, uintptr_t lo, uintptr_t hi) {
    if (g >= lo && g + 16 <= hi) {
        __builtin_nontemporal_store(*reinterpret_cast<const u32x4 *>(chunk), reinterpret_cast<u32x4 *>(g));
        return;
    }
    const uintptr_t s = g > lo ? g : lo, e = g + 16 < hi ? g + 16 : hi; // [s, e) non-empty, < 16 bytes
    const unsigned o = (unsigned)(s - g), n = (unsigned)(e - s);
    const unsigned head = (4u - (o & 3u)) & 3u, hb = head < n ? head : n;
    uint8_t *dst = reinterpret_cast<uint8_t *>(s);
    const uint8_t *src = chunk + o;
    if (hb > 0) dst[0] = src[0];
    if (hb > 1) dst[1] = src[1];
    if (hb > 2) dst[2] = src[2];
    const unsigned body = n - hb, ndw = body >> 2, tb = body & 3;
    const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src + hb); // 4-byte aligned in LDS and in global
    uint32_t *d32 = reinterpret_cast<uint32_t *>(dst + hb);
    if (ndw > 0) d32[0] = s32[0];
    if (ndw > 1) d32[1] = s32[1];
    if (ndw > 2) d32[2] = s32[2];
    const uint8_t *ts = src + hb + 4 * ndw;
    uint8_t *td = dst + hb + 4 * ndw;
    if (tb > 0) td[0] = ts[0];
    if (tb > 1) td[1] = ts[1];
    if (tb > 2) td[2] = ts[2];
}

// value of a 64-bit per-lane quantity in one (wave-uniform) lane, as a scalar: v_readlane_b32 x2
// instead of the ds_bpermute pair a generic __shfl costs
__device__ __forceinline__ unsigned long long read_lane_u64(unsigned long long x, unsigned src_lane) {
    const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)x, (int)src_lane);
    const uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), (int)src_lane);
    return ((unsigned long long)hi << 32) | lo;
}

// read index and word-in-read of word wb + lane without a per-lane 64-bit division: one
// wave-uniform division for the tile's first word, then a 32-bit step for the lane offset
// (t = j0 + lane < wpr + 64): multiply-high by the host's magic = ceil(2^32 / wpr) when
// wpr <= 64, a single compare when wpr > 64.
struct ReadPos { unsigned long long r; unsigned j; };
__device__ __forceinline__ ReadPos fixed_read_pos(unsigned long long wb, unsigned lane, unsigned wpr, unsigned magic, unsigned long long magic64) {
    // wave-uniform floor(wb / wpr) as a multiply-high by the host's floor(2^64 / wpr) + 1 (exact while wb * wpr < 2^64;
    // the host passes 0 for wpr == 1, where that constant would be 2^64)
    const unsigned long long r0 = magic64 ? (unsigned long long)(((unsigned __int128)wb * magic64) >> 64) : wb;
    const unsigned j0 = (unsigned)(wb - r0 * wpr);
    const unsigned t = j0 + lane;
    const unsigned q = wpr > 64 ? (t >= wpr ? 1u : 0u) : (wpr == 1 ? t : __umulhi(t, magic)); // wpr == 1: magic would be 2^32
    return ReadPos{r0 + q, t - q * wpr};
}

// (TileRec, block_owner_kernel: csrc/evidence/batch_evidence.h, evidence build only)


// ---------------------------------------------------------------------------------
// word offsets: exclusive scan of ceil(len_i / 32) -- and, fused into its last pass, the layout plan
// ---------------------------------------------------------------------------------
// Three launches: per-workgroup sums, a one-workgroup scan of those sums, and the pass that writes the offsets.  A thread
// owns kScanPer = 8 CONSECUTIVE sequences: their nine offsets are four 16-byte loads plus one 8-byte load (the lanes of a
// wave read 4 KiB contiguous), the serial part of the scan runs in registers, and the eight results leave as four
// 16-byte stores.  (Round 1's version staged 16 counts per thread through LDS at a stride of 16 dwords: 10-13 bank-conflict
// cycles per LDS instruction and 59 us for 6.7 M sequences, profiles/r02_batch_pmc1.txt.)  A layout plan's last pass is
// plan_emit_kernel below, which redoes this pass's workgroup scan and writes the offsets AND the plan.
constexpr int kScanPer = 8;
constexpr int kScanTile = kBlock * kScanPer; // sequences per workgroup

__device__ __forceinline__ unsigned long long block_exclusive_scan(unsigned long long v, unsigned long long *total) {
    // exclusive scan of one value per thread across the workgroup
    __shared__ unsigned long long wsum[kBlock / 64];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(inc, off);
        if (lane >= (unsigned)off) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned long long before = 0, all = 0;
    for (int i = 0; i < kBlock / 64; ++i) {
        if ((unsigned)i < wave) before += wsum[i];
        all += wsum[i];
    }
    __syncthreads();
    *total = all;
    return before + inc - v;
}

// o[j] = offsets[min(i + j, count)], j = 0..8: entries past the table repeat its last one (sequences of length 0)
__device__ __forceinline__ void load_offsets9(const unsigned long long *__restrict__ offsets, unsigned long long count, unsigned long long i,
                                              unsigned long long (&o)[kScanPer + 1]) {
    if (i + kScanPer <= count) {
        const u32x4_u *src = reinterpret_cast<const u32x4_u *>(offsets + i); // any 8-byte aligned table (gfx950 unaligned-access mode)
#pragma unroll
        for (int q = 0; q < kScanPer / 2; ++q) {
            const u32x4 v = src[q];
            o[2 * q] = ((unsigned long long)v.y << 32) | v.x;
            o[2 * q + 1] = ((unsigned long long)v.w << 32) | v.z;
        }
        o[kScanPer] = offsets[i + kScanPer];
    } else {
#pragma unroll
        for (int j = 0; j <= kScanPer; ++j) o[j] = offsets[i + j < count ? i + j : count];
    }
}

__global__ void __launch_bounds__(kBlock)
word_offsets_block_sums(const unsigned long long *__restrict__ offsets, unsigned long long count,
                        unsigned long long *__restrict__ block_sums) {
    const unsigned long long i = (unsigned long long)blockIdx.x * kScanTile + (unsigned long long)threadIdx.x * kScanPer;
    unsigned long long o[kScanPer + 1], s = 0;
    load_offsets9(offsets, count, i, o);
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) s += (o[j + 1] - o[j] + 31) >> 5;
    unsigned long long total;
    block_exclusive_scan(s, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// one workgroup: exclusive scan of the block sums in place; block_sums[nblocks .. nblocks+2] = the grand total, offsets[0],
// offsets[count] (what the host needs to size a plan: one 24-byte copy)
__global__ void __launch_bounds__(kBlock)
word_offsets_scan_sums(unsigned long long *__restrict__ block_sums, unsigned long long nblocks, const unsigned long long *__restrict__ offsets,
                       unsigned long long count) {
    const unsigned long long per = (nblocks + kBlock - 1) / kBlock;
    const unsigned long long b0 = threadIdx.x * per < nblocks ? threadIdx.x * per : nblocks, b1 = b0 + per < nblocks ? b0 + per : nblocks;
    unsigned long long s = 0;
    for (unsigned long long b = b0; b < b1; ++b) s += block_sums[b];
    unsigned long long total;
    unsigned long long run = block_exclusive_scan(s, &total);
    for (unsigned long long b = b0; b < b1; ++b) {
        const unsigned long long v = block_sums[b];
        block_sums[b] = run;
        run += v;
    }
    if (threadIdx.x == 0) {
        block_sums[nblocks] = total;
        block_sums[nblocks + 1] = offsets[0];
        block_sums[nblocks + 2] = offsets[count];
    }
}

// w[j] -> word_offsets[i + j] for the entries of the table this thread owns (j = 0..7, and the grand total at index `count` by the
// thread that holds the last sequence): four 16-byte stores when the thread's eight sequences are all inside the table
__device__ __forceinline__ void store_word_offsets9(unsigned long long *__restrict__ word_offsets, unsigned long long count, unsigned long long i,
                                                    const unsigned long long (&w)[kScanPer + 1]) {
    if (i + kScanPer <= count) {
        u32x4_u *dst = reinterpret_cast<u32x4_u *>(word_offsets + i);
#pragma unroll
        for (int q = 0; q < kScanPer / 2; ++q)
            dst[q] = u32x4{(uint32_t)w[2 * q], (uint32_t)(w[2 * q] >> 32), (uint32_t)w[2 * q + 1], (uint32_t)(w[2 * q + 1] >> 32)};
        if (i + kScanPer == count) word_offsets[count] = w[kScanPer]; // the grand total, by the thread that holds the last sequence
    } else {
#pragma unroll
        for (int j = 0; j <= kScanPer; ++j)
            if (i + j <= count) word_offsets[i + j] = w[j]; // j with i + j == count: the grand total
    }
}

__global__ void __launch_bounds__(kBlock)
word_offsets_finish(const unsigned long long *__restrict__ offsets, unsigned long long count,
                    const unsigned long long *__restrict__ block_sums, unsigned long long *__restrict__ word_offsets) {
    const unsigned long long i = (unsigned long long)blockIdx.x * kScanTile + (unsigned long long)threadIdx.x * kScanPer;
    unsigned long long o[kScanPer + 1], c[kScanPer], s = 0;
    load_offsets9(offsets, count, i, o);
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) {
        c[j] = (o[j + 1] - o[j] + 31) >> 5;
        s += c[j];
    }
    unsigned long long total;
    unsigned long long run = block_sums[blockIdx.x] + block_exclusive_scan(s, &total);
    unsigned long long w[kScanPer + 1];
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) {
        w[j] = run;
        run += c[j];
    }
    w[kScanPer] = run;
    store_word_offsets9(word_offsets, count, i, w);
}

// ---------------------------------------------------------------------------------
// the layout plan in one pass over the offsets table
// ---------------------------------------------------------------------------------
// bitnuc_encode_batch_dev / bitnuc_decode_batch_dev receive offsets, word offsets and total_words: the plan's sizes are known to
// the host, so the plan can be emitted into context scratch by ONE launch with no host synchronisation and no memset -- and the
// plan kernels run unchanged.  (Round 2's table-driven form searched the owner of every tile in a pre-kernel and rebuilt the
// per-word lookup inside the main kernels: 0.245 / 0.252 ms against 0.205 / 0.206 for the plan kernels.)
// A workgroup owns kScanTile = 2048 consecutive sequences, a thread eight of them.  It needs their word offsets -- and reads
// only ONE of them from the caller's table: word_offsets[i] is by definition the prefix sum of ceil(len/32), so the workgroup's
// first entry plus a workgroup-wide exclusive scan of the counts (registers, wave shuffles, one LDS hop) gives all 2048.  The
// kernel therefore reads 8 bytes per sequence, not 16 (the first form read both tables: 144 MB moved, 24 us; this one 91 MB).
// FROM_TABLE = false: the workgroup's base comes from the scan's block sums instead and WRITE_WO stores the word offsets too --
// the last pass of bitnuc_batch_plan_build_dev (it replaces a pass that scattered the plan bytes into global memory: 44.6 us).
// The workgroup's pad bytes P[W0 + 1 .. W1] are one contiguous run: when it fits kEmitLds bytes (reads: 2048 x 5 words = 10 KiB) it
// is ASSEMBLED IN LDS -- zeroed, pad bytes and tile bases scattered into it with LDS stores -- and leaves as coalesced 16-byte
// stores (every P byte in 1..total_words is written exactly once: no memset); longer runs (sequences of hundreds of bases:
// far fewer table entries per byte) take a per-thread path, ranges above kEmitCoop bytes zeroed by the whole wave.
// bounds[0..1] = offsets[0], offsets[count] for the plan encode's buffer clipping.
constexpr unsigned kEmitCoop = 256;

__device__ __forceinline__ void plan_emit_tile_bases(unsigned long long a, unsigned long long b, unsigned long long o, bool valid,
                                                     unsigned long long *__restrict__ tile_base) {
    const unsigned lane = threadIdx.x & 63;
    const bool has_words = valid && b > a;
    const unsigned long long t0 = (a + 63) >> 6, t1 = has_words ? (b + 63) >> 6 : t0; // tile boundaries 64 t inside [a, b)
    const bool is_long = t1 - t0 > 8;
    if (!is_long)
        for (unsigned long long t = t0; t < t1; ++t) tile_base[t] = o + (((t << 6) - a) << 5);
    unsigned long long m = __ballot(is_long);
    while (m) {
        const unsigned l = (unsigned)__builtin_ctzll(m);
        m &= m - 1;
        const unsigned long long A = read_lane_u64(a, l), O = read_lane_u64(o, l), T0 = read_lane_u64(t0, l), T1 = read_lane_u64(t1, l);
        for (unsigned long long t = T0 + lane; t < T1; t += 64) tile_base[t] = O + (((t << 6) - A) << 5);
    }
}

// zero P[lo .. lo + n): 16-byte stores at any byte address (gfx950 unaligned-access mode), then 8 / 4 / 2 / 1
__device__ __forceinline__ void zero_bytes(uint8_t *__restrict__ p, unsigned long long n) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    unsigned long long k = 0;
    for (; k + 16 <= n; k += 16) *reinterpret_cast<u32x4_u *>(p + k) = z;
    if (n & 8) { *reinterpret_cast<u32x2_u *>(p + k) = u32x2{0u, 0u}; k += 8; }
    if (n & 4) { *reinterpret_cast<u32_u *>(p + k) = 0u; k += 4; }
    if (n & 2) { p[k] = 0; p[k + 1] = 0; k += 2; }
    if (n & 1) p[k] = 0;
}

constexpr unsigned kEmitLds = 16 * 1024;
constexpr unsigned kEmitLdsTiles = kEmitLds / 64 + 2;

template <bool FROM_TABLE, bool WRITE_WO>
__global__ void __launch_bounds__(kBlock)
plan_emit_kernel(const unsigned long long *__restrict__ offsets, const unsigned long long *__restrict__ wbase /* FROM_TABLE: the caller's word offsets; else the exclusive block sums */,
                 unsigned long long count, uint8_t *__restrict__ P, unsigned long long *__restrict__ tile_base, unsigned long long *__restrict__ bounds,
                 unsigned long long *__restrict__ wo_out /* WRITE_WO */) {
    __shared__ __attribute__((aligned(16))) uint8_t pads[kEmitLds + 16];
    __shared__ unsigned long long tbs[kEmitLdsTiles];
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long first = (unsigned long long)blockIdx.x * kScanTile; // < count: the grid is ceil(count / kScanTile)
    const unsigned long long i = first + (unsigned long long)threadIdx.x * kScanPer;
    unsigned long long o[kScanPer + 1], w[kScanPer + 1], s = 0;
    const unsigned long long ic = i < count ? i : count; // threads past the table hold nine copies of its last entry: nothing to write
    load_offsets9(offsets, count, ic, o);
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) s += (o[j + 1] - o[j] + 31) >> 5;
    unsigned long long R; // words of this workgroup's sequences = its pad bytes P[W0 + 1 .. W0 + R]
    const unsigned long long excl = block_exclusive_scan(s, &R);
    const unsigned long long W0 = FROM_TABLE ? wbase[first] : wbase[blockIdx.x]; // wave-uniform: one scalar load
    w[0] = W0 + excl;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) w[j + 1] = w[j] + ((o[j + 1] - o[j] + 31) >> 5);
    if (i == 0) { bounds[0] = o[0]; bounds[1] = offsets[count]; }
    if constexpr (WRITE_WO) if (i <= count) store_word_offsets9(wo_out, count, i, w);
    if (R <= kEmitLds) { // workgroup-uniform
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (unsigned k = 16u * threadIdx.x; k < (unsigned)R; k += 16u * kBlock) *reinterpret_cast<u32x4 *>(pads + k) = z;
        __syncthreads();
        const unsigned long long T0 = (W0 + 63) >> 6; // first tile whose first word lies in [W0, W0 + R)
#pragma unroll
        for (int j = 0; j < kScanPer; ++j) {
            const unsigned long long a = w[j], b = w[j + 1];
            if (ic + j < count && b > a) {
                pads[(unsigned)(b - W0 - 1)] = (uint8_t)(32ull * (b - a) - (o[j + 1] - o[j]));
                for (unsigned long long t = (a + 63) >> 6; t < ((b + 63) >> 6); ++t) tbs[(unsigned)(t - T0)] = o[j] + (((t << 6) - a) << 5);
            }
        }
        __syncthreads();
        uint8_t *dst = P + W0 + 1;
        const unsigned body = (unsigned)R & ~15u;
        for (unsigned k = 16u * threadIdx.x; k < body; k += 16u * kBlock) *reinterpret_cast<u32x4_u *>(dst + k) = *reinterpret_cast<const u32x4 *>(pads + k);
        if (threadIdx.x < (unsigned)R - body) dst[body + threadIdx.x] = pads[body + threadIdx.x];
        const unsigned nt = (unsigned)(((W0 + R + 63) >> 6) - T0);
        for (unsigned k = threadIdx.x; k < nt; k += kBlock) tile_base[T0 + k] = tbs[k];
        return;
    }
    // long ranges: every thread zeroes and fills its own pad bytes P[w[0] + 1 .. w[8]] in global memory
    const unsigned long long len = w[kScanPer] - w[0];
    uint8_t *mine = P + w[0] + 1;
    const bool coop = len > kEmitCoop;
    if (!coop) zero_bytes(mine, len);
    unsigned long long m = __ballot(coop);
    if (m) { // wave-uniform: some lane holds a long range (long sequences) -- the whole wave zeroes it
        while (m) {
            const unsigned l = (unsigned)__builtin_ctzll(m);
            m &= m - 1;
            const unsigned long long W = read_lane_u64(w[0], l), L = read_lane_u64(len, l);
            uint8_t *q = P + W + 1;
            const unsigned long long body = L & ~15ull;
            const u32x4 z = {0u, 0u, 0u, 0u};
            for (unsigned long long k = 16ull * lane; k < body; k += 1024) *reinterpret_cast<u32x4_u *>(q + k) = z;
            if (lane < (unsigned)(L - body)) q[body + lane] = 0;
        }
        // the pad bytes below land inside ranges other lanes have just zeroed: wait until those stores have been performed
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // rare path (sequences of > 8 K bases): its cost does not matter
    }
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) {
        const bool valid = ic + j < count;
        if (valid && w[j + 1] > w[j]) P[w[j + 1]] = (uint8_t)(32ull * (w[j + 1] - w[j]) - (o[j + 1] - o[j]));
        plan_emit_tile_bases(w[j], w[j + 1], o[j], valid, tile_base);
    }
}

// ---- the wave-private LDS strip of 32-bit chunk words (codes of 16 bases), shared by every tile kernel below ----------
// Chunk c lives at strip_slot(c): EVEN chunks at strip[0..], ODD chunks at strip[kSplitOdd..].  A word is 2 chunks long, so
// with a linear strip lane l's three funnel dwords (encode) or three OR targets (decode) start 2 dwords after lane l-1's:
// the 32 lanes of an LDS access group cover 64 dwords = 2 per bank, one extra cycle per group and access (the 6 conflict
// cycles per tile of profiles/r02_batch_pmc1.txt).  Split by parity, a lane of the encode reads even[e], even[e+1],
// odd[e], odd[e+1] with e = byte_off >> 5, which advances by 0 or 1 from lane to lane: at most 32 consecutive indices
// per group, one per bank (equal indices are a broadcast), for every mix of word lengths.  kSplitOdd = 16 mod 32 makes the
// chunk-ordered accesses (fill in the encode, drain in the decode) conflict-free too: a group's 16 even chunks sit in
// banks i..i+15, its 16 odd chunks in banks i+16..i+31.  Measured (profiles/r02_lds_split_strip.txt): SQ_LDS_BANK_CONFLICT
// of the plan encode 3.1 M -> 0 per launch, time unchanged (-0.7 %): the conflicts were real and never the bottleneck.
constexpr int kSplitOdd = 80;    // dword offset of the odd half: >= 68 even entries (chunks 0..135), = 16 mod 32
constexpr int kSplitStrip = 152; // dwords per wave
__device__ __forceinline__ uint32_t *strip_slot(uint32_t *strip, unsigned chunk) { return strip + (chunk >> 1) + (chunk & 1u) * kSplitOdd; }
__device__ __forceinline__ const uint32_t *strip_slot(const uint32_t *strip, unsigned chunk) { return strip + (chunk >> 1) + (chunk & 1u) * kSplitOdd; }

// ---------------------------------------------------------------------------------
// batched encode
// ---------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------
// tile core for CONTIGUOUS layouts (ragged batches, back-to-back fixed-length reads)
// ---------------------------------------------------------------------------------
// When the sequences sit back to back, every byte between the first and the last word of a
// tile is a base of some word of the tile, so the tile is a bulk encode whose 2-bit stream is
// cut at word starts (the kmer_dense_kernel idea with a per-lane cut position): the wave
// encodes the tile's 16-byte chunks (coalesced dwordx4 loads, enc16) into an LDS strip of u32
// code words, and each lane funnel-shifts its 64 bits out of strip dwords (byte_off >> 4) .. +2
// and masks them to 2*nb bits.  Per word that is 3 LDS dword reads and 2 v_alignbit instead of
// 9 LDS reads, 8 v_alignbyte and 8 per-dword packs, and a quarter of the LDS bytes.
// `base`/`nb` = this lane's word (byte offset of its first base, bases in it); lanes past the
// tile's last word mirror it.  span = [span_lo, span_hi) bytes of the tile (wave-uniform).
// strip[c] = codes of chunk c (already in registers: chunk lane+64r in v[r]); validates every
// chunk byte that lies inside the buffer.  Wave-private; call between two wave_lds_fence()s.
__device__ __forceinline__ void stream_fill(const u32x4 (&v)[3], unsigned nchunk, uintptr_t lo16, const uint8_t *__restrict__ seq,
                                            unsigned long long seq_end, uint32_t *strip, unsigned long long *__restrict__ slot,
                                            unsigned long long seq_begin = 0) {
    const unsigned lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const unsigned c = lane + 64 * r;
        if (c < nchunk) {
            uint32_t bad = 0;
            strip_slot(strip, lane)[32 * r] = enc16(v[r], bad); // chunk lane + 64 r: same parity as chunk `lane`, 32 dwords further
            if (__builtin_expect(residue_is_bad(bad), 0)) {
                // a 16-byte aligned chunk may stick out of the buffer at either end: look only at bytes inside it
                const uintptr_t g = lo16 + 16 * (uintptr_t)c, s0 = reinterpret_cast<uintptr_t>(seq), e0 = s0 + seq_end, b0 = s0 + seq_begin;
                const uintptr_t a = g > b0 ? g : b0, b = g + 16 < e0 ? g + 16 : e0;
                if (b > a) rescan_bytes(seq, (unsigned long long)(a - s0), (unsigned)(b - a), slot);
            }
        }
    }
    // (the funnel may read up to 2 chunks past the last one: whatever it finds there is masked off by stream_cut)
}

// the word whose first base sits byte_off bytes into the strip's span, nb bases long (nb = 1..32)
__device__ __forceinline__ unsigned long long stream_cut(const uint32_t *strip, unsigned byte_off, unsigned nb) {
    const unsigned e = byte_off >> 5, sh = (byte_off & 15) * 2;
    const bool odd = (byte_off & 16u) != 0u;
    const uint32_t a0 = strip[e], a1 = strip[e + 1], b0 = strip[kSplitOdd + e], b1 = strip[kSplitOdd + e + 1];
    const uint32_t w0 = odd ? b0 : a0, w1 = odd ? a1 : b0, w2 = odd ? b1 : a1;
    const uint32_t wlo = __builtin_amdgcn_alignbit(w1, w0, sh), whi = __builtin_amdgcn_alignbit(w2, w1, sh);
    const unsigned long long keep = ~0ull >> ((64u - 2u * nb) & 63u); // one 64-bit shift instead of shift / not / compare / select
    return (((unsigned long long)whi << 32) | wlo) & keep;
}

// load + fill + cut for a tile whose span [span_lo, span_hi) is known up front
__device__ __forceinline__ unsigned long long
encode_word_from_stream(const uint8_t *__restrict__ seq, unsigned long long seq_end, unsigned long long span_lo,
                        unsigned long long span_hi, unsigned long long base, unsigned nb, uint32_t *strip /* >= kSplitStrip dwords */,
                        unsigned long long *__restrict__ slot) {
    const unsigned lane = threadIdx.x & 63;
    // read 32 bytes past the last word (clipped to the buffer) so that the 64-bit funnel of a
    // partial last word sees real codes
    unsigned long long hi_off = span_hi + 32;
    if (hi_off > seq_end) hi_off = seq_end;
    const uintptr_t lo = reinterpret_cast<uintptr_t>(seq) + span_lo, hi = reinterpret_cast<uintptr_t>(seq) + hi_off;
    const uintptr_t lo16 = lo & ~(uintptr_t)15;
    const unsigned nchunk = (unsigned)((hi - lo16 + 15) >> 4); // <= 132
    u32x4 v[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const unsigned c = lane + 64 * r;
        v[r] = c < nchunk ? __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(lo16 + 16 * (uintptr_t)c)) : u32x4{0, 0, 0, 0};
    }
    if (span_lo == 0 && lane == 0) v[0] = mask_lead_bytes(v[0], (unsigned)(lo - lo16)); // bytes before the buffer (see mask_lead_bytes)
    wave_lds_fence(); // previous trip's strip readers are done
    stream_fill(v, nchunk, lo16, seq, seq_end, strip, slot);
    wave_lds_fence();
    return stream_cut(strip, (unsigned)(reinterpret_cast<uintptr_t>(seq) + base - lo16), nb);
}

struct BatchLds { // one per wave: the staged byte span of a tile (fixed-length kernels)
    __attribute__((aligned(16))) uint8_t stage[kBatchStage];
};
// ---------------------------------------------------------------------------------
// fixed-length reads: `count` reads of `read_len` bases, read r at byte r*stride
// ---------------------------------------------------------------------------------
// The common sequencing layout (every read the same length, back to back or with a separator).
// Word w belongs to read w / wpr (wpr = ceil(read_len/32)): the lookup is one integer division,
// no offsets tables, no pre-kernel.  Same wave-private tile as encode_batch_kernel: 64 words
// per wave, the tile's byte span staged through LDS with coalesced 16-byte loads when it fits
// (stride - read_len small), per-lane aligned-dword loads from global memory otherwise.
// GAPS (stride > read_len): the bytes after a read's last base are separators, so the validity
// residue of each dword is masked to the bytes that belong to the read.
template <bool GAPS>
__global__ void __launch_bounds__(kBlock)
encode_fixed_kernel(const uint8_t *__restrict__ seq, unsigned read_len, unsigned long long stride, unsigned wpr, unsigned magic, unsigned long long magic64,
                    unsigned long long total_words, unsigned long long seq_end /* bytes in the buffer */, int use_stream,
                    unsigned long long *__restrict__ out, unsigned long long *__restrict__ slot) {
    __shared__ BatchLds lds[kBatchWaves];
    BatchLds &my = lds[wave_in_block()];
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long ntiles = (total_words + kBatchTile - 1) / kBatchTile;
    for (unsigned long long tile = (unsigned long long)blockIdx.x * kBatchWaves + wave_in_block(); tile < ntiles;
         tile += (unsigned long long)gridDim.x * kBatchWaves) {
        const unsigned long long wb = tile * kBatchTile, w = wb + lane;
        const bool active = w < total_words;
        const unsigned last = (unsigned)((total_words - wb < kBatchTile ? total_words - wb : kBatchTile) - 1);
        const ReadPos pos = fixed_read_pos(wb, lane < last ? lane : last, wpr, magic, magic64); // inactive lanes mirror the last word
        const unsigned j = pos.j;
        const unsigned long long base = pos.r * stride + 32ull * j;
        const unsigned left = read_len - 32 * j, nb = left < 32 ? left : 32u;
        // stage 32 bytes past the last word too (clipped to the buffer): the unconditional 32-byte
        // pack of a partial last word then sees the next read's real bytes, not stale LDS
        unsigned long long span_hi = read_lane_u64(base + nb, last) + 32;
        if (span_hi > seq_end) span_hi = seq_end;
        const unsigned long long span_lo = read_lane_u64(base, 0);
        const uintptr_t lo = reinterpret_cast<uintptr_t>(seq) + span_lo, hi = reinterpret_cast<uintptr_t>(seq) + span_hi;
        const uintptr_t lo16 = lo & ~(uintptr_t)15;
        if (!GAPS && use_stream) { // back-to-back reads: cut the tile's 2-bit stream at the word starts
            const unsigned long long word = encode_word_from_stream(seq, seq_end, span_lo, read_lane_u64(base + nb, last), base, nb,
                                                                    reinterpret_cast<uint32_t *>(my.stage), slot);
            if (active) store_packed(word, out + w);
            continue;
        }
        const bool staged = hi - lo16 <= (uintptr_t)(kBatchStage - 16); // wave-uniform
        uint32_t a[9];
        unsigned sh;
        if (staged) {
            const unsigned nchunk = (unsigned)((hi - lo16 + 15) >> 4);
            wave_lds_fence(); // previous trip's LDS readers are done
            for (unsigned c = lane; c < nchunk; c += 64)
                *reinterpret_cast<u32x4 *>(my.stage + 16 * c) = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(lo16 + 16 * (uintptr_t)c));
            wave_lds_fence();
            const unsigned off = (unsigned)(reinterpret_cast<uintptr_t>(seq) + base - lo16);
            sh = off & 3;
            const uint32_t *src = reinterpret_cast<const uint32_t *>(my.stage + (off & ~3u));
#pragma unroll
            for (int i = 0; i < 9; ++i) a[i] = src[i];
        } else { // wide separators: the aligned dwords that hold this word's bytes, straight from global memory
            const uintptr_t p = reinterpret_cast<uintptr_t>(seq) + base;
            sh = (unsigned)(p & 3);
            const unsigned nd = (sh + nb + 3) >> 2;
            const uint32_t *src = reinterpret_cast<const uint32_t *>(p & ~(uintptr_t)3);
#pragma unroll
            for (int i = 0; i < 9; ++i) a[i] = (unsigned)i < nd ? src[i] : 0x41414141u;
        }
        if (!active) continue;
        uint32_t bad = 0, wlo = 0, whi = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t b = 0;
            const uint32_t c = enc4(__builtin_amdgcn_alignbyte(a[i + 1], a[i], sh), b);
            if constexpr (GAPS) { // only this read's bytes may raise the residue
                const int remain = (int)nb - 4 * i;
                const uint32_t m = remain >= 4 ? 0xFFFFFFFFu : (remain <= 0 ? 0u : ((1u << (8 * remain)) - 1u));
                bad |= b & m;
            } else {
                bad |= b;
            }
            if (i < 4) wlo |= c << (8 * i); else whi |= c << (8 * (i - 4));
        }
        const unsigned long long keep = nb >= 32 ? ~0ull : ((1ull << (2 * nb)) - 1);
        wlo &= (uint32_t)keep;
        whi &= (uint32_t)(keep >> 32);
        store_packed(((unsigned long long)whi << 32) | wlo, out + w);
        if (__builtin_expect(residue_is_bad(bad), 0)) rescan_bytes(seq, base, nb, slot);
    }
}

// decode of fixed-length reads; read r's bases go to out[r*stride .. r*stride + read_len).
// CONTIG (stride == read_len): the tile's output bytes are one contiguous span, assembled in LDS
// and written with coalesced 16-byte stores (edges byte-wise).  Otherwise the bytes between reads
// belong to the caller: every lane stores its own 1..32 bytes straight to global memory.
template <bool CONTIG>
__global__ void __launch_bounds__(kBlock)
decode_fixed_kernel(const unsigned long long *__restrict__ words, unsigned read_len, unsigned long long stride, unsigned wpr,
                    unsigned magic, unsigned long long magic64, unsigned long long total_words, uint8_t *__restrict__ out) {
    __shared__ BatchLds lds[kBatchWaves];
    BatchLds &my = lds[wave_in_block()];
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long ntiles = (total_words + kBatchTile - 1) / kBatchTile;
    for (unsigned long long tile = (unsigned long long)blockIdx.x * kBatchWaves + wave_in_block(); tile < ntiles;
         tile += (unsigned long long)gridDim.x * kBatchWaves) {
        const unsigned long long wb = tile * kBatchTile, w = wb + lane;
        const bool active = w < total_words;
        const unsigned last = (unsigned)((total_words - wb < kBatchTile ? total_words - wb : kBatchTile) - 1);
        const unsigned long long word = __builtin_nontemporal_load(words + wb + (lane < last ? lane : last));
        const ReadPos pos = fixed_read_pos(wb, lane < last ? lane : last, wpr, magic, magic64);
        const unsigned j = pos.j;
        const unsigned long long base = pos.r * stride + 32ull * j;
        const unsigned left = read_len - 32 * j, nb = left < 32 ? left : 32u;
        const u32x4 da = dec16((uint32_t)word), db = dec16((uint32_t)(word >> 32));
        const uint32_t d[9] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w, 0u};
        if constexpr (CONTIG) {
            const unsigned long long span_lo = read_lane_u64(base, 0), span_hi = read_lane_u64(base + nb, last);
            const uintptr_t lo = reinterpret_cast<uintptr_t>(out) + span_lo, hi = reinterpret_cast<uintptr_t>(out) + span_hi;
            const uintptr_t lo16 = lo & ~(uintptr_t)15;
            wave_lds_fence();
            if (active) {
                const unsigned off = (unsigned)(reinterpret_cast<uintptr_t>(out) + base - lo16);
                const unsigned head = (4u - (off & 3u)) & 3u;
                const unsigned hb = head < nb ? head : nb;
                const unsigned body = nb - hb, ndw = body >> 2, tb = body & 3;
                if (hb > 0) my.stage[off] = (uint8_t)d[0];
                if (hb > 1) my.stage[off + 1] = (uint8_t)(d[0] >> 8);
                if (hb > 2) my.stage[off + 2] = (uint8_t)(d[0] >> 16);
                uint32_t *dst = reinterpret_cast<uint32_t *>(my.stage + off + hb);
#pragma unroll
                for (int m = 0; m < 8; ++m)
                    if ((unsigned)m < ndw) dst[m] = __builtin_amdgcn_alignbyte(d[m + 1], d[m], head);
                const unsigned q = hb + 4 * ndw;
                const uint32_t tv = dec4((uint32_t)(word >> (2 * q)) & 0xFFu);
                uint8_t *tp = my.stage + off + q;
                if (tb > 0) tp[0] = (uint8_t)tv;
                if (tb > 1) tp[1] = (uint8_t)(tv >> 8);
                if (tb > 2) tp[2] = (uint8_t)(tv >> 16);
            }
            wave_lds_fence();
            const unsigned nchunk = (unsigned)((hi - lo16 + 15) >> 4);
            for (unsigned c = lane; c < nchunk; c += 64)
                store_stage_chunk(my.stage + 16 * c, lo16 + 16 * (uintptr_t)c, lo, hi);
        } else if (active) {
            uint8_t *dst = out + base;
            const unsigned ndw = nb >> 2, tb = nb & 3;
#pragma unroll
            for (int m = 0; m < 8; ++m)
                if ((unsigned)m < ndw) *reinterpret_cast<u32_u *>(dst + 4 * m) = d[m];
            const uint32_t tv = dec4((uint32_t)(word >> (8 * ndw)) & 0xFFu); // bases 4*ndw ..
            if (tb > 0) dst[4 * ndw] = (uint8_t)tv;
            if (tb > 1) dst[4 * ndw + 1] = (uint8_t)(tv >> 8);
            if (tb > 2) dst[4 * ndw + 2] = (uint8_t)(tv >> 16);
        }
    }
}


// ---------------------------------------------------------------------------------
// bit-strip decode of back-to-back fixed-length reads (stride == read_len)
// ---------------------------------------------------------------------------------
// The tile's output is one contiguous byte run and 16 bytes of it are exactly 32 bits of the run's 2-bit stream.
// So the wave first rebuilds that stream, pad bits squeezed out, in an LDS strip indexed from the run's 16-byte
// aligned start: lane = one word, masked to its 2*nb bits and OR-ed in at bit 2*(byte offset) with three
// ds_or_b32 (the strip is zeroed first).  Strip dword c then IS output chunk c: a lane reads one dword, decodes 16
// bases and issues one coalesced dwordx4 store -- the store pattern of the bulk decode, no byte staging and no
// predicated LDS stores.  Only the run's first / last chunk (shared with the neighbouring tiles) is written
// byte-wise, through the wave's stage buffer and store_stage_chunk.
constexpr int kStripDwords = kBatchTile * 2 + 8; // 64 words x 64 bits + the <= 15-byte lead + slack (the decode strip is LINEAR: see strip_or_word)

// the three steps of a bit-strip tile, shared by decode_fixed_strip_kernel and decode_batch_kernel (wave-private)
__device__ __forceinline__ void strip_zero(uint32_t *strip, unsigned lane) {
#pragma unroll
    for (int j = 0; j < (kStripDwords + 63) / 64; ++j)
        if (lane + 64 * j < (unsigned)kStripDwords) strip[lane + 64 * j] = 0u;
}
// OR the low 2*nb bits of `word` into the strip at bit offset `bit` (even, < 32 * (kStripDwords - 2))
__device__ __forceinline__ void strip_or_word(uint32_t *strip, unsigned bit, unsigned long long word, unsigned nb) {
    const unsigned drop = 64u - 2u * nb;                      // 0 for a full word: both shifts are then the identity
    const unsigned long long v = (word << drop) >> drop;      // bits above 2*nb are ignored, like the reference's decode
    const unsigned sh = bit & 31;
    const unsigned long long t = v << sh; // bits 0..63 of the 96-bit shifted value; the rest is the third dword
    uint32_t *dst = strip + (bit >> 5);
    atomicOr(dst, (uint32_t)t);
    atomicOr(dst + 1, (uint32_t)(t >> 32));
    atomicOr(dst + 2, ((uint32_t)(v >> 32) >> 1) >> (31u - sh)); // == (v >> 32) >> (32 - sh), and 0 for sh == 0, without a select
}
// strip dword c is the 16-byte chunk at lo16 + 16c: decode and store it.  The run's first / last chunk is shared with the
// neighbouring tiles when lo / hi are not 16-byte aligned.  Those two edges are NOT written byte by byte (that cost ~30
// memory instructions per tile for two lanes' worth of data): the lane that holds the partial first chunk stores the run's
// FIRST 16 bytes [lo, lo+16), the lane that holds the partial last chunk its LAST 16 bytes [hi-16, hi), as one unaligned
// dwordx4 each.  They overlap this tile's own aligned chunks with identical bytes and never touch a neighbour's.  Every lane
// takes its 32 code bits from the strip at a BIT position (two dword reads + v_alignbit; shift 0 for the whole chunks), so
// the two edges ride in the same pass as the whole chunks -- a separate pass for them cost a second whole-wave dec16 per
// tile -- and which chunks are whole / on a shared line is decided by comparing the 32-bit chunk index with wave-uniform
// bounds, not by 64-bit address arithmetic per lane.  Runs shorter than 16 bytes (the last tile of a small batch) take the
// byte-wise path.
// POLICY: cache policy of the whole-chunk stores.  0 = streaming (nt), 1 = allocating (plain), 2 (shipped) = plain for the chunks
// of the run's first / last 128-byte cache line (shared with the neighbouring tiles, so that the two parts of a line can meet in
// L2), nt for everything else (inline asm: written as two C++ stores the compiler merges them into one plain flat store).
// Sustained bursts, 150-base / 1000-base reads (profiles/r02_plan_decode_store_policy.txt): 2 = 5.86 / 6.19 TB/s,
// 0 = 5.79 / 6.07, 1 = 5.38 / 5.73.
template <int POLICY = 2>
__device__ __forceinline__ void strip_drain(const uint32_t *strip, uint8_t (*edge)[16], uint8_t *__restrict__ out, uintptr_t lo16, uintptr_t lo, uintptr_t hi, unsigned lane) {
    const unsigned nchunk = (unsigned)((hi - lo16 + 15) >> 4); // <= 130
    const uintptr_t op = reinterpret_cast<uintptr_t>(out);
    if (hi - lo >= 16) { // wave-uniform
        const unsigned lead = (unsigned)(lo - lo16), tail = (unsigned)(hi - lo16); // bytes from lo16: first base, one past the last
        const unsigned c0 = lead ? 1u : 0u, c1 = tail >> 4;                        // whole chunks: c0 <= c < c1
        // chunks on the run's first / last 128-byte line: c < cl0, c >= cl1
        const unsigned cl0 = (unsigned)((((lo | 127) + 1) - lo16) >> 4);
        const uintptr_t last_line = (hi - 1) & ~(uintptr_t)127;
        const unsigned cl1 = last_line > lo16 ? (unsigned)((last_line - lo16) >> 4) : 0u;
        uint8_t *base = out + (lo16 - op); // derived from `out`: global (not flat) stores
        if (lead == 0 && (tail & 15u) == 0) { // wave-uniform: a run of whole chunks only (e.g. 100-base reads: a tile is 16 reads = 1600 bytes)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned c = lane + 64 * j;
                if (c >= nchunk) break;
                const u32x4 d = dec16(strip[c]);
                uint8_t *dst = base + 16u * c;
                bool plain = POLICY == 1;
                if constexpr (POLICY == 2) plain = c < cl0 || c >= cl1;
                if (plain) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(d) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(d) : "memory");
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const unsigned c = lane + 64 * j;
            if (c >= nchunk) break;
            const bool whole = c >= c0 && c < c1;
            const unsigned boff = whole ? 16u * c : (c == 0 ? lead : tail - 16u); // first byte (from lo16) of the 16 this lane stores
            const unsigned bit = 2u * boff;
            const uint32_t w0 = strip[bit >> 5], w1 = strip[(bit >> 5) + 1];
            const u32x4 d = dec16(__builtin_amdgcn_alignbit(w1, w0, bit & 31));
            uint8_t *dst = base + boff;
            bool plain = !whole || POLICY == 1;
            if constexpr (POLICY == 2) plain = plain || c < cl0 || c >= cl1;
            if (plain) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(d) : "memory"); // any byte address (unaligned-access mode)
            else asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(d) : "memory");
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) { // a run shorter than 16 bytes: byte-wise through the wave's stage buffer
        const unsigned c = lane + 64 * j;
        if (c >= nchunk) break;
        const uintptr_t g = lo16 + 16 * (uintptr_t)c;
        uint8_t *e = edge[c ? 1 : 0];
        *reinterpret_cast<u32x4 *>(e) = dec16(strip[c]);
        store_stage_chunk(e, g, lo, hi);
    }
}

// (decode_fixed_strip_kernel: csrc/evidence/batch_evidence.h, evidence build only)


// (round 2's table-driven kernels -- tile records + O(1) pad-scatter lookup -- : csrc/evidence/batch_evidence.h, evidence build only)

// ---- helpers shared by the plan kernels below -------------------------------------------------------------------------------
constexpr int kB2Strip = kSplitStrip; // dwords of the code / bit strip: chunk 0..128 + funnel slack, in the split layout

// lane i <- lane i-1 across the wave (gfx9 DPP wave_shr:1); lane 0 gets 0
__device__ __forceinline__ uint32_t lane_shr1(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
}

// inclusive prefix sum over the 64 lanes: row_shr 1, 2, 4, 8 inside each row of 16, then row_bcast:15 into rows 1 and 3
// and row_bcast:31 into rows 2 and 3 (six DPP adds, no LDS)
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /* row_bcast:31 */, 0xC, 0xF, false);
    return v;
}

// residue of chunk c (16 bytes at lo16 + 16 c) flagged: latch the first invalid byte among those of the chunk that
// lie inside the batch [seq_begin, seq_end)
__device__ __forceinline__ void rescan_chunk(const uint8_t *__restrict__ seq, uintptr_t lo16, unsigned c, unsigned long long seq_begin,
                                             unsigned long long seq_end, unsigned long long *__restrict__ slot) {
    const uintptr_t s0 = reinterpret_cast<uintptr_t>(seq), g = lo16 + 16 * (uintptr_t)c;
    const uintptr_t a = g > s0 + seq_begin ? g : s0 + seq_begin, b = g + 16 < s0 + seq_end ? g + 16 : s0 + seq_end;
    if (b > a) rescan_bytes(seq, (unsigned long long)(a - s0), (unsigned)(b - a), slot);
}

// =================================================================================
// ragged batches with a layout PLAN (bitnuc_batch_plan): the shipped fast path
// =================================================================================
// What the ablations of the table-driven kernels showed (tools/ab_batch_ablate.py, profiles/r02_batch_ablation.txt): with
// the window loads and the tile-record pre-kernel taken out, the ragged kernels run at the speed of the fixed-length
// ones.  Both are properties of the LAYOUT, not of the data, and a layout is used at least twice (encode, later decode)
// and often many times.  The plan therefore holds, per layout:
//   tile_base[t]  byte offset of word 64 t (one u64 per wave tile), and
//   P[w]          one byte per word: the padding of the sequence that ENDS at word w-1 (0..31; 0 when word w does
//                 not start a sequence) -- written by plan_emit_kernel.
// A lane's lookup is then ONE byte load, issued together with the tile's data, and a DPP prefix sum:
//   n  = P[wb + lane + 1]          what word (wb + lane) lacks to 32 bases,
//   nb = 32 - n,    first byte = tile_base + 32 lane - (sum of n over the lanes before it).
// No window, no search, no LDS for the lookup, no dependent global load, no per-call pre-kernel.
// (The plan is written by plan_emit_kernel above.)

// One tile of the plan encode.  issue: the tile's 16-byte chunks (2 per lane), one dword of the 129th chunk (an unaligned
// tile's last <= 15 bytes; lanes 0-3) and the lane's pad byte.  Every load is unconditional with a clamped, in-bounds
// address (a lane past the tile re-reads its last chunk): four memory instructions per tile on every path.  What a clamped
// lane loads is never looked at: stream_cut masks everything past a word's bases.
struct PlanEncTile {
    u32x4 v0, v1;
    uint32_t x2, n;
};
struct PlanEncGeom { // wave-uniform
    unsigned long long wb;
    long long off16; // byte offset (from seq) of the 16-byte aligned chunk that holds the tile's first base; may be -15..-1
    unsigned lead;   // the first base's offset inside that chunk
    unsigned nchunk; // 1..129
    unsigned last;   // index of the tile's last real word
};
__device__ __forceinline__ PlanEncGeom plan_enc_geom(const uint8_t *seq, unsigned long long tile, unsigned long long base0, unsigned long long total_words,
                                                     unsigned long long seq_end) {
    PlanEncGeom g;
    g.wb = tile * kBatchTile;
    g.last = (unsigned)((total_words - g.wb < kBatchTile ? total_words - g.wb : kBatchTile) - 1);
    g.lead = (unsigned)((reinterpret_cast<uintptr_t>(seq) + base0) & 15);
    g.off16 = (long long)base0 - (long long)g.lead;
    const unsigned long long hi = base0 + kBatchTile * 32 < seq_end ? base0 + kBatchTile * 32 : seq_end; // offset past the tile's last byte
    g.nchunk = (long long)hi > g.off16 ? (unsigned)(((long long)hi - g.off16 + 15) >> 4) : 1u;         // a tile holds >= 1 base
    return g;
}
// ABL (evidence build, tools/ab_plan_enc_ablate.py; right only for batches of 32-base reads, whose tiles are dense and aligned):
// bit 0 = tile base by arithmetic instead of the tile_base load, bit 1 = no pad-byte load, bit 2 = no load of the 129th chunk.
template <int ABL = 0>
__device__ __forceinline__ void plan_enc_issue(const uint8_t *__restrict__ seq, const uint8_t *__restrict__ P, const PlanEncGeom &g, unsigned lane, PlanEncTile &t) {
    const u32x4 *src = reinterpret_cast<const u32x4 *>(seq + g.off16); // derived from the kernel argument: global_load, not flat_load
    const unsigned top = g.nchunk - 1;
    if constexpr (ABL & 2) t.n = 0u;
    else t.n = (uint32_t)P[g.wb + (lane < g.last ? lane : g.last) + 1]; // the whole lookup: one byte
    t.v0 = __builtin_nontemporal_load(src + (lane < top ? lane : top));
    t.v1 = __builtin_nontemporal_load(src + (lane + 64 < top ? lane + 64 : top));
    if constexpr (ABL & 4) t.x2 = 0x41414141u;
    else t.x2 = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(src + (top < 128u ? top : 128u)) + (lane & 3));
}
__device__ __forceinline__ void plan_enc_finish(const uint8_t *__restrict__ seq, const PlanEncGeom &g, const PlanEncTile &t, unsigned long long base0,
                                                unsigned long long seq_begin, unsigned long long seq_end, uint32_t *strip, unsigned lane,
                                                unsigned long long *__restrict__ out, unsigned long long *__restrict__ slot) {
    const uintptr_t lo16 = reinterpret_cast<uintptr_t>(seq) + (uintptr_t)g.off16;
    const uint32_t n = lane <= g.last ? t.n : 0u;
    // Only the slow path uses x2, and the compiler otherwise SINKS its load into that path, behind the wait for v0 / v1: a
    // second, dependent memory round trip per unaligned tile (0.228 ms instead of 0.209 ms on 150-base reads).  Pinning
    // the value here costs nothing: it is the youngest of four loads that are waited for together.
    uint32_t x2 = t.x2;
    asm volatile("" : "+v"(x2));
    if (__ballot(n != 0u) == 0ull && g.last == 63u && g.lead == 0 && g.nchunk >= 128u) {
        // fast tile (wave-uniform): 64 full words, first base 16-byte aligned = a plain 2 KiB bulk encode
        uint32_t bad = 0;
        uint32_t *o32 = reinterpret_cast<uint32_t *>(out + g.wb);
        const uint32_t c0 = enc16(t.v0, bad), c1 = enc16(t.v1, bad);
        store_packed(c0, o32 + lane);
        store_packed(c1, o32 + 64 + lane);
        if (__builtin_expect(residue_is_bad(bad), 0)) {
            rescan_bytes(seq, base0 + 16 * lane, 16, slot);
            rescan_bytes(seq, base0 + 16 * (lane + 64), 16, slot);
        }
        return;
    }
    uint32_t b0 = 0, b1 = 0, b2 = 0;
    u32x4 v0 = t.v0;
    if (g.wb == 0 && g.lead != 0 && lane == 0) v0 = mask_lead_bytes(v0, g.lead); // the batch's first chunk: bytes before its first base are not its own
    const uint32_t c0 = enc16(v0, b0), c1 = enc16(t.v1, b1), c2 = enc4(x2, b2);
    wave_lds_fence(); // the previous tile's strip readers are done
    uint32_t *mine = strip_slot(strip, lane); // chunk `lane`; chunk lane + 64 has the same parity: 32 dwords further
    mine[0] = c0;
    mine[32] = c1;
    if (lane < 4) reinterpret_cast<uint8_t *>(strip_slot(strip, 128))[lane] = (uint8_t)c2; // chunk 128's codes, a byte per lane
    if (__builtin_expect(residue_is_bad(b0 | b1 | b2), 0)) {
        if (residue_is_bad(b0) && lane < g.nchunk) rescan_chunk(seq, lo16, lane, seq_begin, seq_end, slot);
        if (residue_is_bad(b1) && lane + 64 < g.nchunk) rescan_chunk(seq, lo16, lane + 64, seq_begin, seq_end, slot);
        if (residue_is_bad(b2) && lane < 4 && g.nchunk > 128) rescan_chunk(seq, lo16, 128, seq_begin, seq_end, slot);
    }
    const unsigned base_rel = 32u * lane - (wave_inclusive_sum(n) - n), nb = 32u - n;
    wave_lds_fence();
    if (lane <= g.last) {
        const unsigned long long word = stream_cut(strip, g.lead + base_rel, nb);
        store_packed(word, out + g.wb + lane);
    }
}

// One tile per wave trip; the grid normally covers every tile and the hardware dispatcher walks them.  (Resident "walking"
// waves that keep the next 1..3 tiles' loads and the tile base in flight were built on this issue / finish split and
// measured 3-12 % SLOWER at every grid size: profiles/r02_plan_encode_walking_waves.txt.)
// U = consecutive tiles per wave trip: their bases come from one scalar load and all their chunk loads are issued before the
// first tile is encoded (tile_base[t] -> data address is a dependent pair of round trips; U tiles share it).
template <int U, int ABL = 0>
__global__ void __launch_bounds__(kBlock)
encode_batch_plan_kernel(const uint8_t *__restrict__ seq, const unsigned long long *__restrict__ tile_base, const uint8_t *__restrict__ P,
                         unsigned long long total_words, const unsigned long long *__restrict__ bounds /* offsets[0], offsets[count]: device memory */,
                         unsigned long long *__restrict__ out, unsigned long long *__restrict__ slot) {
    __shared__ uint32_t strips[kBatchWaves][kB2Strip];
    const unsigned long long seq_begin = bounds[0], seq_end = bounds[1]; // wave-uniform (scalar loads), issued with the first tile base
    const unsigned lane = threadIdx.x & 63, wave = wave_in_block();
    uint32_t *strip = strips[wave];
    const unsigned long long ntiles = (total_words + kBatchTile - 1) / kBatchTile;
    const unsigned waves = blockDim.x >> 6; // <= kBatchWaves
    for (unsigned long long t0 = ((unsigned long long)blockIdx.x * waves + wave) * U; t0 < ntiles; t0 += (unsigned long long)gridDim.x * waves * U) {
        unsigned long long base0[U];
        PlanEncGeom g[U];
        PlanEncTile t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned long long tile = t0 + u < ntiles ? t0 + u : ntiles - 1; // clamp: redundant but in bounds
            if constexpr (ABL & 1) base0[u] = seq_begin + tile * (kBatchTile * 32ull);
            else base0[u] = tile_base[tile];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            g[u] = plan_enc_geom(seq, t0 + u < ntiles ? t0 + u : ntiles - 1, base0[u], total_words, seq_end);
            plan_enc_issue<ABL>(seq, P, g[u], lane, t[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (t0 + u >= ntiles) break; // wave-uniform
            plan_enc_finish(seq, g[u], t[u], base0[u], seq_begin, seq_end, strip, lane, out, slot);
        }
    }
}

// One decode tile through the bit strip, shared by the plan and the fixed-length kernels.  base0 = byte offset (in out) of the
// tile's first base; base_rel / nb = this lane's word relative to it; dense = no word of the tile is short (wave-uniform).
template <int POLICY>
__device__ __forceinline__ void decode_tile_strip(uint32_t *strip, uint8_t (*edge)[16], uint8_t *__restrict__ out, unsigned long long base0, unsigned base_rel,
                                                  unsigned nb, bool dense, unsigned last, unsigned long long word, unsigned lane) {
    const uintptr_t op = reinterpret_cast<uintptr_t>(out), lo = op + base0, lo16 = lo & ~(uintptr_t)15;
    if (dense && last == 63u && (lo & 15) == 0) {
        // fast tile (wave-uniform): the words cross the strip once so that lane l owns 16-base groups l and l+64
        wave_lds_fence();
        reinterpret_cast<unsigned long long *>(strip)[lane] = word;
        wave_lds_fence();
        const uint32_t h0 = strip[lane], h1 = strip[64 + lane];
        uint8_t *dst = out + base0;
        store_group<true, true>(dst + 16 * lane, dec16(h0));
        store_group<true, true>(dst + 16 * (lane + 64), dec16(h1));
        return;
    }
    wave_lds_fence(); // the previous tile's strip readers are done
    strip_zero(strip, lane);
    const unsigned end_rel = (unsigned)__builtin_amdgcn_readlane((int)(base_rel + nb), (int)last);
    const uintptr_t hi = lo + end_rel;
    wave_lds_fence();
    if (lane <= last) strip_or_word(strip, 2u * ((unsigned)(lo - lo16) + base_rel), word, nb);
    wave_lds_fence();
    if (hi > lo) strip_drain<POLICY>(strip, edge, out, lo16, lo, hi, lane);
}

// U = tiles per wave trip: the loads of U consecutive tiles (word, pad byte, tile base) are issued before the first is
// processed, and the line shared by two of the wave's own tiles is written by one wave.
template <int POLICY, int U>
__global__ void __launch_bounds__(kBlock)
decode_batch_plan_kernel(const unsigned long long *__restrict__ words, const unsigned long long *__restrict__ tile_base, const uint8_t *__restrict__ P,
                         unsigned long long total_words, uint8_t *__restrict__ out) {
    __shared__ uint32_t strips[kBatchWaves][kB2Strip];
    __shared__ __attribute__((aligned(16))) uint8_t edge[kBatchWaves][2][16];
    const unsigned lane = threadIdx.x & 63, wave = wave_in_block();
    uint32_t *strip = strips[wave];
    const unsigned long long ntiles = (total_words + kBatchTile - 1) / kBatchTile;
    for (unsigned long long t0 = ((unsigned long long)blockIdx.x * kBatchWaves + wave) * U; t0 < ntiles;
         t0 += (unsigned long long)gridDim.x * kBatchWaves * U) {
        unsigned long long word[U], base0[U];
        uint32_t n[U];
        unsigned last[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned long long tile = t0 + u < ntiles ? t0 + u : ntiles - 1; // clamp: redundant but in bounds
            const unsigned long long wb = tile * kBatchTile;
            last[u] = (unsigned)((total_words - wb < kBatchTile ? total_words - wb : kBatchTile) - 1);
            word[u] = __builtin_nontemporal_load(words + wb + (lane < last[u] ? lane : last[u]));
            n[u] = lane <= last[u] ? (uint32_t)P[wb + lane + 1] : 0u;
            base0[u] = tile_base[tile];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (t0 + u >= ntiles) break; // wave-uniform
            const bool dense = __ballot(n[u] != 0u) == 0ull;
            const unsigned base_rel = 32u * lane - (wave_inclusive_sum(n[u]) - n[u]), nb = 32u - n[u];
            decode_tile_strip<POLICY>(strip, edge[wave], out, base0[u], base_rel, nb, dense, last[u], word[u], lane);
        }
    }
}

// (decode_batch_plan_lines_kernel, round 4's line- / chunk-owning plan decode: csrc/evidence/batch_evidence.h, evidence build only)

// Back-to-back fixed-length reads through the same tile body: the plan's two lookups are arithmetic here.  Wave-uniform
// (scalar unit): r0 = floor(wb / wpr) by the host's multiply-high constant, j0 = wb - r0 wpr, base0 = r0 read_len + 32 j0.
// Per lane, all 32-bit and relative to the tile: t = j0 + lane, q = floor(t / wpr) (t < wpr + 64), j = t - q wpr,
// first byte = q read_len + 32 (j - j0), bases = min(32, read_len - 32 j).
template <int POLICY>
__global__ void __launch_bounds__(kBlock)
decode_fixed_tile_kernel(const unsigned long long *__restrict__ words, unsigned read_len, unsigned wpr, unsigned magic, unsigned long long magic64,
                         unsigned long long total_words, uint8_t *__restrict__ out) {
    __shared__ uint32_t strips[kBatchWaves][kB2Strip];
    __shared__ __attribute__((aligned(16))) uint8_t edge[kBatchWaves][2][16];
    const unsigned lane = threadIdx.x & 63, wave = wave_in_block();
    uint32_t *strip = strips[wave];
    const unsigned long long ntiles = (total_words + kBatchTile - 1) / kBatchTile;
    const bool dense = (read_len & 31u) == 0u;
    for (unsigned long long tile = (unsigned long long)blockIdx.x * kBatchWaves + wave; tile < ntiles; tile += (unsigned long long)gridDim.x * kBatchWaves) {
        const unsigned long long wb = tile * kBatchTile;
        const unsigned last = (unsigned)((total_words - wb < kBatchTile ? total_words - wb : kBatchTile) - 1);
        const unsigned l = lane < last ? lane : last; // lanes past the tile mirror its last word
        const unsigned long long word = __builtin_nontemporal_load(words + wb + l);
        const unsigned long long r0 = magic64 ? (unsigned long long)(((unsigned __int128)wb * magic64) >> 64) : wb;
        const unsigned j0 = (unsigned)(wb - r0 * wpr);
        const unsigned long long base0 = r0 * read_len + 32ull * j0;
        const unsigned t = j0 + l;
        const unsigned q = wpr > 64 ? (t >= wpr ? 1u : 0u) : (wpr == 1 ? t : __umulhi(t, magic));
        const unsigned j = t - q * wpr;
        const unsigned base_rel = q * read_len + 32u * j - 32u * j0;
        const unsigned left = read_len - 32u * j, nb = left < 32u ? left : 32u;
        decode_tile_strip<POLICY>(strip, edge[wave], out, base0, base_rel, nb, dense, last, word, lane);
    }
}

#ifdef BITNUC_SWEEP_VARIANTS
#include "evidence/batch_evidence.h" // the formulations that lost their A/B: evidence build only
#endif

} // namespace bitnuc_dev
