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
//   * word -> sequence lookup: a small pre-kernel finds the owner of every tile's first word
//     (one binary search per tile, all in parallel), then each lane searches the wave's LDS
//     window of the next offsets (global fallback if a run of empty sequences overflows it).
#pragma once
#include "codec_device.h"

namespace bitnuc_dev {

constexpr int kBatchTile = 64;                      // words per WAVE: every wave works alone on its own tile, so the
                                                    // kernels have no workgroup barrier at all
constexpr int kBatchWin = 128;                      // offsets window per wave (>= kBatchTile + 1, + slack for empty sequences)
constexpr int kBatchStage = kBatchTile * 32 + 128;  // 2 KiB span + alignment / read-ahead slack, per wave
constexpr int kBatchWaves = kBlock / 64;            // waves (= tiles in flight) per workgroup

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

struct WordLoc {
    unsigned long long base; // byte offset of the word's first base in the sequence buffer
    unsigned nb;             // bases in this word (1..32)
};

// rec[b] = {sequence that owns word 64*b, byte offset of that word's first base}: one thread
// per wave tile, so the ~log2(count) dependent loads of the search are paid once, in
// parallel, instead of by every wave -- and the main kernels can start fetching a tile's
// bytes straight after reading its 16-byte record.
struct TileRec { unsigned long long owner, base0, avail; }; // avail = bases left in the owner sequence from base0

template <int EST>
__global__ void __launch_bounds__(kBlock)
block_owner_kernel(const unsigned long long *__restrict__ offsets, const unsigned long long *__restrict__ word_offsets,
                   unsigned long long count, unsigned long long total_words, unsigned long long ntiles,
                   unsigned long long ratio64 /* floor(2^64 * count / total_words), EST == 2 */, TileRec *__restrict__ recs) {
    const unsigned long long b = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    if (b >= ntiles) return;
    const unsigned long long wb = b * kBatchTile;
    // owner = upper_bound(word_offsets[0..count], wb) - 1.  Interpolate first (exact for
    // equal-length reads), gallop outwards to bracket, then bisect the bracket: a handful of
    // dependent loads instead of log2(count).
    unsigned long long lo, hi; // invariant: word_offsets[lo] <= wb < word_offsets[hi]
    {
        unsigned long long est;
        if constexpr (EST == 0) est = (unsigned long long)(((unsigned __int128)wb * count) / total_words);
        else if constexpr (EST == 1) est = (unsigned long long)((double)wb * ((double)count / (double)total_words));
        else { // floor(wb * count / total_words) exactly, without a 128-bit division: the 0.64 fixed-point product is the
               // floor or one below it, and the remainder (exact in wrapping 64-bit arithmetic: it is < 2 * total_words) tells which
            est = (unsigned long long)(((unsigned __int128)wb * ratio64) >> 64);
            const unsigned long long rem = wb * count - est * total_words;
            if (rem >= total_words) ++est;
        }
        unsigned long long p = est < count ? est : count - 1, step = 1;
        if (word_offsets[p] <= wb) {
            lo = p;
            hi = p + 1;
            while (hi < count && word_offsets[hi] <= wb) { lo = hi; hi = hi + step < count ? hi + step : count; step <<= 1; }
            // word_offsets[count] = total_words > wb closes the bracket
        } else {
            hi = p;
            lo = p > 0 ? p - 1 : 0;
            while (lo > 0 && word_offsets[lo] > wb) { hi = lo; lo = lo > step ? lo - step : 0; step <<= 1; }
        }
    }
    while (hi - lo > 1) {
        const unsigned long long mid = (lo + hi) >> 1;
        if (word_offsets[mid] <= wb) lo = mid; else hi = mid;
    }
    const unsigned long long base0 = offsets[lo] + ((wb - word_offsets[lo]) << 5);
    recs[b] = TileRec{lo, base0, offsets[lo + 1] - base0};
}

// Resolve this lane's word from the wave's LDS window (already holding `filled` = 64 entries
// starting at sequence sb).  If the tile's last word is not covered yet the window grows 64
// sequences at a time (150-base reads need 14 entries; 1-word sequences need 65).
// Wave-private: only wave-level fences, no workgroup barrier.
__device__ __forceinline__ WordLoc locate_word(const unsigned long long *__restrict__ offsets,
                                               const unsigned long long *__restrict__ word_offsets,
                                               unsigned long long count, unsigned long long sb, unsigned long long wb,
                                               unsigned long long w, bool active, unsigned long long *win_wo,
                                               unsigned long long *win_so, unsigned hi0) {
    const unsigned lane = threadIdx.x & 63;
    unsigned filled = 64;
    while (filled <= (unsigned)kBatchWin && win_wo[filled - 1] <= wb + kBatchTile - 1) { // wave-uniform, rare
        const unsigned i = filled + lane;
        if (i <= (unsigned)kBatchWin) {
            const unsigned long long s = sb + i < count ? sb + i : count;
            win_wo[i] = word_offsets[s];
            win_so[i] = offsets[s];
        }
        filled = filled + 64 <= (unsigned)kBatchWin + 1 ? filled + 64 : kBatchWin + 1;
        wave_lds_fence();
    }
    WordLoc loc{0, 0};
    if (!active) return loc;
    // upper_bound in the window.  Entry 0 (the tile's owner) starts at or before every word of the
    // tile, and the caller knows the first entry that starts past the tile (hi0, from one ballot over
    // the entries while they were still in registers): the search runs over the handful of
    // sequences that really start inside the tile, not over all 64 entries.
    unsigned lo = 1, hi = filled == 64 ? hi0 : filled;
    while (lo < hi) {
        const unsigned mid = (lo + hi) >> 1;
        if (win_wo[mid] <= w) lo = mid + 1; else hi = mid;
    }
    unsigned long long w0, s0, s1;
    if (lo < filled) {
        w0 = win_wo[lo - 1]; s0 = win_so[lo - 1]; s1 = win_so[lo];
    } else { // more sequences start inside this tile than the window holds
        const unsigned long long s = owner_of_word(word_offsets, count, w);
        w0 = word_offsets[s]; s0 = offsets[s]; s1 = offsets[s + 1];
    }
    loc.base = s0 + ((w - w0) << 5);
    const unsigned long long left = s1 - loc.base;
    loc.nb = left < 32 ? (unsigned)left : 32u;
    return loc;
}

// index of the first of the wave's 64 window entries that starts past the tile's last word (64 if none)
__device__ __forceinline__ unsigned first_entry_past(unsigned long long entry, unsigned long long last_word) {
    const unsigned long long m = __ballot(entry > last_word);
    return m ? (unsigned)__builtin_ctzll(m) : 64u;
}

// Tile-relative lookup for the common case that the wave's 64 window entries reach past the tile (hi0 < 64).
// Entry i is kept as two 32-bit numbers relative to the tile: its first word minus the tile's first word, and its
// first byte minus base0 (the byte of the tile's first base); the owner (entry 0) starts at or before the tile and
// is entered as (0, 0).  The search, the byte offset and the length are then 32-bit arithmetic on 32-bit LDS
// entries.  Returns the word's first byte relative to base0 and its base count.
struct RelLoc { unsigned base, nb; };
__device__ __forceinline__ RelLoc locate_word_rel(unsigned long long wo_r, unsigned long long so_r, unsigned long long wb,
                                                  unsigned long long base0, unsigned hi0, uint32_t *win /* 128 entries */) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long dw = wo_r - wb, ds = so_r - base0; // entries >= 1 start after the tile's first word / base
    win[lane] = lane == 0 ? 0u : (dw < 0xFFFFull ? (unsigned)dw : 0xFFFFu);
    win[64 + lane] = lane == 0 ? 0u : (ds < 0x7FFFFFFFull ? (unsigned)ds : 0x7FFFFFFFu);
    wave_lds_fence();
    unsigned lo = 1, hi = hi0; // upper_bound over the entries that start inside the tile
    while (lo < hi) {
        const unsigned mid = (lo + hi) >> 1;
        if (win[mid] <= lane) lo = mid + 1; else hi = mid;
    }
    const unsigned w0 = win[lo - 1], s0 = win[64 + lo - 1], s1 = win[64 + lo];
    RelLoc loc;
    loc.base = s0 + ((lane - w0) << 5);
    const unsigned left = s1 - loc.base;
    loc.nb = left < 32 ? left : 32u;
    return loc;
}

// ---------------------------------------------------------------------------------
// word offsets: exclusive scan of ceil(len_i / 32), three small kernels
// ---------------------------------------------------------------------------------
constexpr int kScanItems = 16; // per thread
__device__ __forceinline__ unsigned long long words_of(const unsigned long long *__restrict__ offsets, unsigned long long i) {
    return (offsets[i + 1] - offsets[i] + 31) >> 5;
}

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

// A workgroup owns kBlock*kScanItems consecutive sequences.  Global accesses are coalesced
// (sequence t + kBlock*j in round j); the per-thread serial part (16 consecutive sequences
// per thread, needed for the scan order) runs on an LDS copy of the word counts.
constexpr int kScanTile = kBlock * kScanItems;

__device__ __forceinline__ void load_counts_to_lds(const unsigned long long *__restrict__ offsets, unsigned long long count,
                                                   unsigned long long i0, uint32_t *cnt /* [kScanTile] */) {
    for (int j = 0; j < kScanItems; ++j) {
        const unsigned long long i = i0 + (unsigned long long)j * kBlock + threadIdx.x;
        // ceil(len/32) of one sequence fits u32 for sequences < 2^37 bases; longer ones saturate the
        // u32 and are re-read exactly in the serial part below
        unsigned long long w = i < count ? words_of(offsets, i) : 0ull;
        cnt[j * kBlock + threadIdx.x] = w > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)w;
    }
    __syncthreads();
}

__device__ __forceinline__ unsigned long long count_at(const unsigned long long *__restrict__ offsets, unsigned long long i,
                                                       const uint32_t *cnt, unsigned local) {
    const uint32_t c = cnt[local];
    return c == 0xFFFFFFFFu ? words_of(offsets, i) : c;
}

__global__ void __launch_bounds__(kBlock)
word_offsets_block_sums(const unsigned long long *__restrict__ offsets, unsigned long long count,
                        unsigned long long *__restrict__ block_sums) {
    __shared__ uint32_t cnt[kScanTile];
    const unsigned long long i0 = (unsigned long long)blockIdx.x * kScanTile;
    load_counts_to_lds(offsets, count, i0, cnt);
    unsigned long long s = 0;
    for (int j = 0; j < kScanItems; ++j) {
        const unsigned local = threadIdx.x * kScanItems + j;
        if (i0 + local < count) s += count_at(offsets, i0 + local, cnt, local);
    }
    unsigned long long total;
    block_exclusive_scan(s, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(kBlock)
word_offsets_scan_sums(unsigned long long *__restrict__ block_sums, unsigned long long nblocks) {
    // single workgroup: thread t owns a contiguous chunk of the block sums
    const unsigned long long per = (nblocks + kBlock - 1) / kBlock;
    const unsigned long long b0 = threadIdx.x * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
    unsigned long long s = 0;
    for (unsigned long long b = b0; b < b1; ++b) s += block_sums[b];
    unsigned long long total;
    unsigned long long run = block_exclusive_scan(s, &total);
    for (unsigned long long b = b0; b < b1; ++b) {
        const unsigned long long v = block_sums[b];
        block_sums[b] = run;
        run += v;
    }
}

__global__ void __launch_bounds__(kBlock)
word_offsets_finish(const unsigned long long *__restrict__ offsets, unsigned long long count,
                    const unsigned long long *__restrict__ block_sums, unsigned long long *__restrict__ word_offsets) {
    __shared__ uint32_t cnt[kScanTile];
    __shared__ unsigned long long res[kScanTile];
    const unsigned long long i0 = (unsigned long long)blockIdx.x * kScanTile;
    load_counts_to_lds(offsets, count, i0, cnt);
    unsigned long long v[kScanItems], s = 0;
    for (int j = 0; j < kScanItems; ++j) {
        const unsigned local = threadIdx.x * kScanItems + j;
        v[j] = i0 + local < count ? count_at(offsets, i0 + local, cnt, local) : 0;
        s += v[j];
    }
    unsigned long long total;
    unsigned long long run = block_sums[blockIdx.x] + block_exclusive_scan(s, &total);
    for (int j = 0; j < kScanItems; ++j) {
        res[threadIdx.x * kScanItems + j] = run;
        run += v[j];
    }
    __syncthreads();
    for (int j = 0; j < kScanItems; ++j) { // coalesced write-out
        const unsigned local = j * kBlock + threadIdx.x;
        if (i0 + local < count) word_offsets[i0 + local] = res[local];
    }
    // total number of words: written by the workgroup that holds the last sequence
    if (threadIdx.x == kBlock - 1 && i0 + kScanTile >= count) word_offsets[count] = block_sums[blockIdx.x] + total;
}

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
            strip[c] = enc16(v[r], bad);
            if (__builtin_expect(residue_is_bad(bad), 0)) {
                // a 16-byte aligned chunk may stick out of the buffer at either end: look only at bytes inside it
                const uintptr_t g = lo16 + 16 * (uintptr_t)c, s0 = reinterpret_cast<uintptr_t>(seq), e0 = s0 + seq_end, b0 = s0 + seq_begin;
                const uintptr_t a = g > b0 ? g : b0, b = g + 16 < e0 ? g + 16 : e0;
                if (b > a) rescan_bytes(seq, (unsigned long long)(a - s0), (unsigned)(b - a), slot);
            }
        }
    }
    if (lane < 4) strip[nchunk + lane] = 0; // the funnel may read up to 2 dwords past the last chunk
}

// the word whose first base sits byte_off bytes into the strip's span, nb bases long
__device__ __forceinline__ unsigned long long stream_cut(const uint32_t *strip, unsigned byte_off, unsigned nb) {
    const unsigned d = byte_off >> 4, sh = (byte_off & 15) * 2;
    const uint32_t w0 = strip[d], w1 = strip[d + 1], w2 = strip[d + 2];
    const uint32_t wlo = __builtin_amdgcn_alignbit(w1, w0, sh), whi = __builtin_amdgcn_alignbit(w2, w1, sh);
    const unsigned long long keep = nb >= 32 ? ~0ull : ((1ull << (2 * nb)) - 1);
    return (((unsigned long long)whi << 32) | wlo) & keep;
}

// load + fill + cut for a tile whose span [span_lo, span_hi) is known up front
__device__ __forceinline__ unsigned long long
encode_word_from_stream(const uint8_t *__restrict__ seq, unsigned long long seq_end, unsigned long long span_lo,
                        unsigned long long span_hi, unsigned long long base, unsigned nb, uint32_t *strip /* >= 136 dwords */,
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
    wave_lds_fence(); // previous trip's strip readers are done
    stream_fill(v, nchunk, lo16, seq, seq_end, strip, slot);
    wave_lds_fence();
    return stream_cut(strip, (unsigned)(reinterpret_cast<uintptr_t>(seq) + base - lo16), nb);
}

struct BatchLds { // one per wave and tile in flight
    unsigned long long win_wo[kBatchWin + 1], win_so[kBatchWin + 1];
    __attribute__((aligned(16))) uint8_t stage[kBatchStage];
};
constexpr int kBatchInFlight = 1; // tiles whose loads a wave issues before it computes on the first

// A wave handles kBatchInFlight consecutive tiles per trip in three phases, so that the
// dependent global loads of a tile (record -> offsets window / bytes) overlap with the other
// tile's instead of adding up: (A) tile records, (B) window entries + the tile's bytes
// (always the 2 KiB after its first base, clipped at the buffer end: known from the record
// alone), (C) LDS lookup, funnel, encode, store.
template <bool STREAM>
__global__ void __launch_bounds__(kBlock)
encode_batch_kernel(const uint8_t *__restrict__ seq, const unsigned long long *__restrict__ offsets, const unsigned long long *__restrict__ word_offsets,
                    unsigned long long count, unsigned long long total_words, const TileRec *__restrict__ recs,
                    unsigned long long *__restrict__ out, unsigned long long *__restrict__ slot) {
    constexpr int U = kBatchInFlight;
    __shared__ BatchLds lds[kBatchWaves][U];
    const unsigned lane = threadIdx.x & 63, wave = wave_in_block();
    const unsigned long long ntiles = (total_words + kBatchTile - 1) / kBatchTile;
    const unsigned long long seq_end = offsets[count]; // end of the sequence buffer: bounds the 2 KiB tile fetch
    const unsigned long long seq_begin = offsets[0];   // bytes before the batch are never examined
    for (unsigned long long t0 = ((unsigned long long)blockIdx.x * kBatchWaves + wave) * U; t0 < ntiles;
         t0 += (unsigned long long)gridDim.x * kBatchWaves * U) {
        // (A) records
        TileRec rec[U];
#pragma unroll
        for (int u = 0; u < U; ++u) rec[u] = recs[t0 + u < ntiles ? t0 + u : ntiles - 1];
        // (B) window entries and bytes, all in flight together
        unsigned long long wo_r[U], so_r[U];
        u32x4 st[U][3];
        uintptr_t lo16[U];
        unsigned nchunk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned long long s = rec[u].owner + lane < count ? rec[u].owner + lane : count;
            wo_r[u] = word_offsets[s];
            so_r[u] = offsets[s];
            const unsigned long long hi_off = rec[u].base0 + kBatchTile * 32 + 32 < seq_end ? rec[u].base0 + kBatchTile * 32 + 32 : seq_end;
            const uintptr_t lo = reinterpret_cast<uintptr_t>(seq) + rec[u].base0, hi = reinterpret_cast<uintptr_t>(seq) + hi_off;
            lo16[u] = lo & ~(uintptr_t)15;
            nchunk[u] = hi > lo16[u] ? (unsigned)((hi - lo16[u] + 15) >> 4) : 0; // <= 129
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const unsigned c = lane + 64 * j;
                st[u][j] = c < nchunk[u] ? __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(lo16[u] + 16 * (uintptr_t)c)) : u32x4{0, 0, 0, 0};
            }
        }
        // (C) per tile
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (t0 + u >= ntiles) break; // wave-uniform
            BatchLds &my = lds[wave][u];
            const unsigned long long wb = (t0 + u) * kBatchTile, w = wb + lane;
            const bool active = w < total_words;
            if (rec[u].avail >= kBatchTile * 32) {
                // fast tile (wave-uniform): all 64 words are full and inside one sequence, so the
                // tile is a plain 2 KiB bulk encode: the bytes already sit in st[u][0..1] as 128
                // coalesced 16-base groups (group g = chunk g when base0 is 16-byte aligned)
                if (((reinterpret_cast<uintptr_t>(seq) + rec[u].base0) & 15) == 0) {
                    uint32_t bad = 0;
                    uint32_t *o32 = reinterpret_cast<uint32_t *>(out + wb);
                    const uint32_t c0 = enc16(st[u][0], bad), c1 = enc16(st[u][1], bad);
                    __builtin_nontemporal_store(c0, o32 + lane);
                    __builtin_nontemporal_store(c1, o32 + 64 + lane);
                    if (__builtin_expect(residue_is_bad(bad), 0)) {
                        rescan_bytes(seq, rec[u].base0 + 16 * lane, 16, slot);
                        rescan_bytes(seq, rec[u].base0 + 16 * (lane + 64), 16, slot);
                    }
                    continue;
                }
            }
            if constexpr (STREAM) {
                // The sequences are back to back, so the tile is a bulk encode whose 2-bit stream is cut at the word
                // starts (stream_fill / stream_cut).  The lookup runs first: it gives the exact end of the tile's last
                // word, so only the chunks that hold the tile's bases are encoded (two rounds of enc16, not three).
                uint32_t *strip = reinterpret_cast<uint32_t *>(my.stage);
                const unsigned hi0 = first_entry_past(wo_r[u], wb + kBatchTile - 1);
                wave_lds_fence(); // previous trip's LDS readers are done
                // the first 2 KiB of chunks are needed by (almost) every tile: encode them while the lookup is in flight
                const unsigned n01 = nchunk[u] < 128 ? nchunk[u] : 128u;
                stream_fill(st[u], n01, lo16[u], seq, seq_end, strip, slot, seq_begin);
                unsigned off, nb;
                unsigned long long base;
                const unsigned lastl = (unsigned)((total_words - wb < kBatchTile ? total_words - wb : kBatchTile) - 1);
                if (hi0 < 64) {
                    const RelLoc rl = locate_word_rel(wo_r[u], so_r[u], wb, rec[u].base0, hi0, reinterpret_cast<uint32_t *>(my.win_wo));
                    nb = rl.nb;
                    base = rec[u].base0 + rl.base;
                } else {
                    my.win_wo[lane] = wo_r[u];
                    my.win_so[lane] = so_r[u];
                    wave_lds_fence();
                    const WordLoc loc = locate_word(offsets, word_offsets, count, rec[u].owner, wb, wb + (lane < lastl ? lane : lastl), true,
                                                    my.win_wo, my.win_so, hi0);
                    nb = loc.nb;
                    base = loc.base;
                }
                off = (unsigned)(reinterpret_cast<uintptr_t>(seq) + base - lo16[u]);
                const unsigned long long span_hi = read_lane_u64(base + nb, lastl);
                const unsigned need = (unsigned)((reinterpret_cast<uintptr_t>(seq) + span_hi - lo16[u] + 15) >> 4);
                if (need > 128) { // wave-uniform and rare: the tile's last bases sit in the third round of chunks (<= 2 of them)
                    const unsigned c = 128 + lane;
                    if (c < need && c < nchunk[u]) {
                        uint32_t bad = 0;
                        strip[c] = enc16(st[u][2], bad);
                        if (__builtin_expect(residue_is_bad(bad), 0)) {
                            const uintptr_t g = lo16[u] + 16 * (uintptr_t)c, s0 = reinterpret_cast<uintptr_t>(seq), e0 = s0 + seq_end;
                            const uintptr_t b = g + 16 < e0 ? g + 16 : e0;
                            if (b > g) rescan_bytes(seq, (unsigned long long)(g - s0), (unsigned)(b - g), slot);
                        }
                    }
                    if (lane < 4) strip[(need < nchunk[u] ? need : nchunk[u]) + lane] = 0; // the funnel may read 2 dwords past the last chunk
                }
                wave_lds_fence();
                if (!active) continue;
                const unsigned long long word = stream_cut(strip, off, nb);
                __builtin_nontemporal_store(word, out + w);
            } else {
                wave_lds_fence(); // previous trip's LDS readers are done
    #pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const unsigned c = lane + 64 * j;
                    if (c < nchunk[u]) *reinterpret_cast<u32x4 *>(my.stage + 16 * c) = st[u][j];
                }
                const unsigned hi0 = first_entry_past(wo_r[u], wb + kBatchTile - 1);
                unsigned off, nb;
                unsigned long long base; // absolute byte offset of the word's first base (error reporting only)
                if (hi0 < 64) { // the window reaches past the tile (always, unless > 63 sequences start inside it)
                    const RelLoc rl = locate_word_rel(wo_r[u], so_r[u], wb, rec[u].base0, hi0, reinterpret_cast<uint32_t *>(my.win_wo));
                    off = (unsigned)(reinterpret_cast<uintptr_t>(seq) + rec[u].base0 - lo16[u]) + rl.base;
                    nb = rl.nb;
                    base = rec[u].base0 + rl.base;
                } else {
                    my.win_wo[lane] = wo_r[u];
                    my.win_so[lane] = so_r[u];
                    wave_lds_fence();
                    const WordLoc loc = locate_word(offsets, word_offsets, count, rec[u].owner, wb, w, active, my.win_wo, my.win_so, hi0);
                    off = (unsigned)(reinterpret_cast<uintptr_t>(seq) + loc.base - lo16[u]);
                    nb = loc.nb;
                    base = loc.base;
                }
                if (!active) continue;
                // The word's bytes start at any byte offset: read the 9 ALIGNED LDS dwords that cover 32
                // bytes from there and funnel-shift (misaligned ds_read_b32 works on gfx950 but runs ~2x
                // slower).  These kernels are VALU-issue bound (PMC), so there is no per-dword length
                // logic: all 32 bytes are packed -- past the word's nb bases they are the next sequence's
                // bytes or stage slack -- and the packed word is masked to 2*nb bits instead.  A residue
                // from those extra bytes only sends the lane to rescan_bytes, which looks at its own nb.
                const unsigned sh = off & 3;
                const uint32_t *src = reinterpret_cast<const uint32_t *>(my.stage + (off & ~3u));
                uint32_t a[9];
    #pragma unroll
                for (int i = 0; i < 9; ++i) a[i] = src[i];
                uint32_t bad = 0, wlo = 0, whi = 0;
    #pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t r = enc4(__builtin_amdgcn_alignbyte(a[i + 1], a[i], sh), bad);
                    if (i < 4) wlo |= r << (8 * i); else whi |= r << (8 * (i - 4));
                }
                const unsigned long long keep = nb >= 32 ? ~0ull : ((1ull << (2 * nb)) - 1);
                wlo &= (uint32_t)keep;
                whi &= (uint32_t)(keep >> 32);
                __builtin_nontemporal_store(((unsigned long long)whi << 32) | wlo, out + w);
                if (__builtin_expect(residue_is_bad(bad), 0)) rescan_bytes(seq, base, nb, slot);
            }
        }
    }
}

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
            if (active) __builtin_nontemporal_store(word, out + w);
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
        __builtin_nontemporal_store(((unsigned long long)whi << 32) | wlo, out + w);
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
constexpr int kStripDwords = kBatchTile * 2 + 8; // 64 words x 64 bits + the <= 15-byte lead + slack

// the three steps of a bit-strip tile, shared by decode_fixed_strip_kernel and decode_batch_kernel (wave-private)
__device__ __forceinline__ void strip_zero(uint32_t *strip, unsigned lane) {
#pragma unroll
    for (int j = 0; j < (kStripDwords + 63) / 64; ++j)
        if (lane + 64 * j < (unsigned)kStripDwords) strip[lane + 64 * j] = 0u;
}
// OR the low 2*nb bits of `word` into the strip at bit offset `bit` (even, < 32 * (kStripDwords - 2))
__device__ __forceinline__ void strip_or_word(uint32_t *strip, unsigned bit, unsigned long long word, unsigned nb) {
    const unsigned long long keep = nb >= 32 ? ~0ull : ((1ull << (2 * nb)) - 1);
    const unsigned long long v = word & keep;
    const unsigned sh = bit & 31;
    const unsigned long long t = v << sh; // bits 0..63 of the 96-bit shifted value; the rest is the third dword
    uint32_t *dst = strip + (bit >> 5);
    atomicOr(dst, (uint32_t)t);
    atomicOr(dst + 1, (uint32_t)(t >> 32));
    atomicOr(dst + 2, sh ? (uint32_t)(v >> 32) >> (32 - sh) : 0u);
}
// strip dword c is the 16-byte chunk at lo16 + 16c: decode and store it; the run's first / last chunk (shared with
// the neighbouring tiles) only as far as [lo, hi) reaches, through a 16-byte LDS slot and store_stage_chunk
__device__ __forceinline__ void strip_drain(const uint32_t *strip, uint8_t (*edge)[16], uintptr_t lo16, uintptr_t lo, uintptr_t hi, unsigned lane) {
    const unsigned nchunk = (unsigned)((hi - lo16 + 15) >> 4); // <= 130
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const unsigned c = lane + 64 * j;
        if (c >= nchunk) break;
        const u32x4 d = dec16(strip[c]);
        const uintptr_t g = lo16 + 16 * (uintptr_t)c;
        if (g >= lo && g + 16 <= hi) {
            __builtin_nontemporal_store(d, reinterpret_cast<u32x4 *>(g));
        } else {
            uint8_t *e = edge[c ? 1 : 0];
            *reinterpret_cast<u32x4 *>(e) = d;
            store_stage_chunk(e, g, lo, hi);
        }
    }
}

__global__ void __launch_bounds__(kBlock)
decode_fixed_strip_kernel(const unsigned long long *__restrict__ words, unsigned read_len, unsigned wpr, unsigned magic, unsigned long long magic64,
                          unsigned long long total_words, uint8_t *__restrict__ out) {
    __shared__ uint32_t strips[kBatchWaves][kStripDwords];
    __shared__ __attribute__((aligned(16))) uint8_t edge[kBatchWaves][2][16];
    uint32_t *strip = strips[wave_in_block()];
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long ntiles = (total_words + kBatchTile - 1) / kBatchTile;
    for (unsigned long long tile = (unsigned long long)blockIdx.x * kBatchWaves + wave_in_block(); tile < ntiles;
         tile += (unsigned long long)gridDim.x * kBatchWaves) {
        const unsigned long long wb = tile * kBatchTile;
        const unsigned last = (unsigned)((total_words - wb < kBatchTile ? total_words - wb : kBatchTile) - 1);
        const bool active = lane <= last;
        const unsigned long long word = __builtin_nontemporal_load(words + wb + (lane < last ? lane : last));
        const ReadPos pos = fixed_read_pos(wb, lane < last ? lane : last, wpr, magic, magic64);
        const unsigned long long base = pos.r * read_len + 32ull * pos.j; // stride == read_len
        const unsigned left = read_len - 32 * pos.j, nb = left < 32 ? left : 32u;
        const unsigned long long span_lo = read_lane_u64(base, 0), span_hi = read_lane_u64(base + nb, last);
        const uintptr_t lo = reinterpret_cast<uintptr_t>(out) + span_lo, hi = reinterpret_cast<uintptr_t>(out) + span_hi;
        const uintptr_t lo16 = lo & ~(uintptr_t)15;
        wave_lds_fence(); // previous trip's readers are done
        strip_zero(strip, lane);
        wave_lds_fence();
        if (active) strip_or_word(strip, 2u * (unsigned)(reinterpret_cast<uintptr_t>(out) + base - lo16), word, nb);
        wave_lds_fence();
        strip_drain(strip, edge[wave_in_block()], lo16, lo, hi, lane);
    }
}

// ---------------------------------------------------------------------------------
// batched decode: sequence i's bases go to out[offsets[i] .. offsets[i+1])
// ---------------------------------------------------------------------------------
// Wave-private, like encode_batch: the tile is the wave's 64 words (same records, same tile-relative lookup), the
// output run is rebuilt as a 2-bit stream in the wave's bit strip (see decode_fixed_strip_kernel) and leaves as
// aligned dwordx4 stores; no workgroup barrier.  (A 128-word workgroup tile with a byte scatter was the faster
// form until the bit strip: profiles/r01_ab_decode_batch_wave_private.txt.)
__global__ void __launch_bounds__(kBlock)
decode_batch_kernel(const unsigned long long *__restrict__ words, const unsigned long long *__restrict__ word_offsets,
                         const unsigned long long *__restrict__ offsets, unsigned long long count,
                         unsigned long long total_words, const TileRec *__restrict__ recs, uint8_t *__restrict__ out) {
    __shared__ uint32_t strips[kBatchWaves][kStripDwords];
    __shared__ unsigned long long wins[kBatchWaves][2 * (kBatchWin + 1)];
    __shared__ __attribute__((aligned(16))) uint8_t edge[kBatchWaves][2][16];
    const unsigned wv = wave_in_block(), lane = threadIdx.x & 63;
    uint32_t *strip = strips[wv];
    unsigned long long *win_wo = wins[wv], *win_so = wins[wv] + kBatchWin + 1;
    const unsigned long long ntiles = (total_words + kBatchTile - 1) / kBatchTile;
    for (unsigned long long tile = (unsigned long long)blockIdx.x * kBatchWaves + wv; tile < ntiles;
         tile += (unsigned long long)gridDim.x * kBatchWaves) {
        const unsigned long long wb = tile * kBatchTile;
        const unsigned last = (unsigned)((total_words - wb < kBatchTile ? total_words - wb : kBatchTile) - 1);
        const bool active = lane <= last;
        const unsigned long long word = __builtin_nontemporal_load(words + wb + (lane < last ? lane : last)); // independent of the lookup
        const TileRec rec = recs[tile];
        if (rec.avail >= kBatchTile * 32 && ((reinterpret_cast<uintptr_t>(out) + rec.base0) & 15) == 0) {
            // fast tile (wave-uniform): 64 full words inside one sequence; the words cross the strip once so that
            // lane l owns 16-base groups l and l+64
            wave_lds_fence();
            reinterpret_cast<unsigned long long *>(strip)[lane] = word;
            wave_lds_fence();
            const uint32_t h0 = strip[lane], h1 = strip[64 + lane];
            uint8_t *dst = out + rec.base0;
            store_group<true, true>(dst + 16 * lane, dec16(h0));
            store_group<true, true>(dst + 16 * (lane + 64), dec16(h1));
            continue;
        }
        const unsigned long long si = rec.owner + lane < count ? rec.owner + lane : count;
        const unsigned long long wo_r = word_offsets[si], so_r = offsets[si];
        const unsigned hi0 = first_entry_past(wo_r, wb + kBatchTile - 1);
        wave_lds_fence(); // previous trip's readers are done
        strip_zero(strip, lane);
        unsigned long long base; // absolute byte offset of the lane's word
        unsigned nb;
        const unsigned long long wq = wb + (lane < last ? lane : last); // inactive lanes mirror the last word
        if (hi0 < 64) {
            const RelLoc rl = locate_word_rel(wo_r, so_r, wb, rec.base0, hi0, reinterpret_cast<uint32_t *>(win_wo));
            // locate_word_rel resolves word wb + lane; an inactive lane's result is never used
            base = rec.base0 + rl.base;
            nb = rl.nb;
        } else {
            win_wo[lane] = wo_r;
            win_so[lane] = so_r;
            wave_lds_fence();
            const WordLoc loc = locate_word(offsets, word_offsets, count, rec.owner, wb, wq, true, win_wo, win_so, hi0);
            base = loc.base;
            nb = loc.nb;
        }
        const unsigned long long span_hi = read_lane_u64(base + nb, last);
        const uintptr_t lo = reinterpret_cast<uintptr_t>(out) + rec.base0, hi = reinterpret_cast<uintptr_t>(out) + span_hi;
        const uintptr_t lo16 = lo & ~(uintptr_t)15;
        wave_lds_fence(); // zeroing (and the lookup's LDS traffic) before the ORs
        if (active) strip_or_word(strip, 2u * (unsigned)(reinterpret_cast<uintptr_t>(out) + base - lo16), word, nb);
        wave_lds_fence();
        strip_drain(strip, edge[wv], lo16, lo, hi, lane);
    }
}

} // namespace bitnuc_dev
