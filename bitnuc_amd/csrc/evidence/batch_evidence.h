// batch_evidence.h -- EVIDENCE BUILD ONLY (-DBITNUC_SWEEP_VARIANTS): read-batch kernels that lost their A/B (profiles/NARRATIVE_r01_r03.md 3.5,
// profiles/README.md).  Included at the end of batch_device.h, inside namespace bitnuc_dev, after everything they share with the shipped kernels.
// Nothing in the product library instantiates or even sees this file.
#pragma once

// ==== TileRec, block_owner_kernel ====

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

// ==== decode_fixed_strip_kernel ====

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
        strip_drain(strip, edge[wave_in_block()], out, lo16, lo, hi, lane);
    }
}

// ==== round 2's table-driven kernels (tile records + O(1) pad-scatter lookup): Batch2Lds, tile_lookup, encode_batch2_kernel, decode_batch2_kernel ====

// =================================================================================
// ragged batches, second formulation (round 2): O(1) word -> sequence lookup, stream-cut encode
// =================================================================================
// What the PMC counters said about the first formulation (profiles/r01_batch_pmc_sq*.txt): the waves sit in
// s_waitcnt (68 %), the LDS array is busy 45 % (encode) / 71 % (decode) of the launch, and the bank-conflict counter
// (in quad-cycles) equals the nine 8-way conflicting funnel reads of the byte stage (lane stride 32 B = 8 dwords).
// So this form removes LDS work, not VALU work:
//   * word -> sequence lookup without a search.  Lane i holds window entry i = sequence (owner + i): its first word
//     and first byte relative to the tile (dw_i, ds_i).  The padding of the sequence before it,
//     pad_{i-1} = 32 (dw_i - dw_{i-1}) - (ds_i - ds_{i-1})  (0..31: what the last word of sequence i-1 lacks),
//     is ADDED into an LDS array at index dw_i (the tile word where sequence i starts; ds_add_u32, so empty
//     sequences, which start at the same word and have pad 0, need no special case).  Lane l then reads pads[l] and
//     pads[l+1]; an inclusive DPP prefix sum P(l) of pads[0..l] is the padding that precedes word l, hence
//         first byte of word l = base0 + 32 l - P(l),      bases in word l = 32 - pads[l+1].
//     One zeroing store, one add, one read2 and six DPP adds replace the window stores, a 4-7 step binary search
//     (dependent LDS round trips) and three entry reads.
//   * encode cuts the tile's 2-bit stream (stream_fill / stream_cut above): the staged data are the enc16 code words
//     (a quarter of the bytes), and a lane's three funnel dwords sit 2 dwords from its neighbour's (2-way at worst)
//     instead of nine reads at an 8-dword stride (8-way).
// The window is 64 entries, a second round of 64 is fetched when more than 63 sequences start inside one tile
constexpr int kB2Pads = 68;   // pads[0..64] + slack

struct Batch2Lds {
    uint32_t strip[kB2Strip];
    uint32_t pads[kB2Pads];
    __attribute__((aligned(16))) uint8_t edge[2][16];
};



// Resolve word (wb + lane) of the tile: *base_rel = its first byte relative to rec.base0, *nb = its bases.
// wo / so = this lane's window entry (first word / first byte of sequence owner + lane, clamped to `count`).
// last_abs = the tile's last real word.  Wave-private; `pads` needs kB2Pads dwords.
__device__ __forceinline__ void tile_lookup(const unsigned long long *__restrict__ offsets, const unsigned long long *__restrict__ word_offsets,
                                            unsigned long long count, const TileRec &rec, unsigned long long wo, unsigned long long so,
                                            unsigned long long wb, unsigned long long last_abs, uint32_t *pads, unsigned &base_rel, unsigned &nb) {
    const unsigned lane = threadIdx.x & 63;
    uint32_t dw = (uint32_t)(wo - wb), ds = (uint32_t)(so - rec.base0); // exact mod 2^32; used only where the true values are small
    bool reaches = __ballot(wo > last_abs) != 0ull;                     // some entry starts past the tile: the window covers it
    wave_lds_fence(); // the previous trip's readers of pads are done
    pads[lane] = 0u;
    if (lane < (unsigned)(kB2Pads - 64)) pads[64 + lane] = 0u;
    wave_lds_fence();
    {
        const uint32_t pad = 32u * (dw - lane_shr1(dw)) - (ds - lane_shr1(ds));
        if (lane >= 1 && wo <= wb + 64) atomicAdd(&pads[dw], pad); // sequence `lane` starts at tile word dw (1..64)
    }
    if (!reaches) { // more than 63 sequences start inside the tile: entries 64..127 (wave-uniform, rare)
        const uint32_t dw63 = (uint32_t)__builtin_amdgcn_readlane((int)dw, 63), ds63 = (uint32_t)__builtin_amdgcn_readlane((int)ds, 63);
        const unsigned long long s2 = rec.owner + 64 + lane < count ? rec.owner + 64 + lane : count;
        const unsigned long long wo2 = word_offsets[s2], so2 = offsets[s2];
        const uint32_t dw2 = (uint32_t)(wo2 - wb), ds2 = (uint32_t)(so2 - rec.base0);
        uint32_t pdw = lane_shr1(dw2), pds = lane_shr1(ds2);
        if (lane == 0) { pdw = dw63; pds = ds63; }
        const uint32_t pad = 32u * (dw2 - pdw) - (ds2 - pds);
        if (wo2 <= wb + 64) atomicAdd(&pads[dw2], pad);
        reaches = __ballot(wo2 > last_abs) != 0ull;
    }
    wave_lds_fence();
    if (__builtin_expect(reaches, 1)) {
        const uint32_t p0 = pads[lane], p1 = pads[lane + 1];
        base_rel = 32u * lane - wave_inclusive_sum(p0);
        nb = 32u - p1;
        return;
    }
    // more sequence starts than two windows hold (runs of empty sequences): per-lane search in global memory
    const unsigned long long w = wb + lane < last_abs ? wb + lane : last_abs;
    const unsigned long long sidx = owner_of_word(word_offsets, count, w);
    const unsigned long long b = offsets[sidx] + ((w - word_offsets[sidx]) << 5), left = offsets[sidx + 1] - b;
    base_rel = (unsigned)(b - rec.base0);
    nb = left < 32 ? (unsigned)left : 32u;
}


// ABL != 0: timing-only ablations for tools/ab_batch_ablate.py (the output is WRONG): bit 0 = no window loads, bit 1 = no
// pad scatter / scan (fixed fake positions), bit 2 = decode: no partial edge chunks, bit 3 = no tile-record load.
template <int ABL>
__global__ void __launch_bounds__(kBlock)
encode_batch2_kernel(const uint8_t *__restrict__ seq, const unsigned long long *__restrict__ offsets, const unsigned long long *__restrict__ word_offsets,
                     unsigned long long count, unsigned long long total_words, const TileRec *__restrict__ recs,
                     unsigned long long *__restrict__ out, unsigned long long *__restrict__ slot) {
    __shared__ Batch2Lds lds[kBatchWaves];
    const unsigned lane = threadIdx.x & 63, wave = wave_in_block();
    Batch2Lds &my = lds[wave];
    const unsigned long long ntiles = (total_words + kBatchTile - 1) / kBatchTile;
    const unsigned long long seq_end = offsets[count], seq_begin = offsets[0];
    const uintptr_t sp = reinterpret_cast<uintptr_t>(seq);
    for (unsigned long long tile = (unsigned long long)blockIdx.x * kBatchWaves + wave; tile < ntiles;
         tile += (unsigned long long)gridDim.x * kBatchWaves) {
        TileRec rec;
        if constexpr (ABL & 8) { rec.owner = tile * 12; rec.base0 = tile * 1900 + offsets[0]; rec.avail = 150; }
        else rec = recs[tile];
        const unsigned long long wb = tile * kBatchTile;
        const unsigned last = (unsigned)((total_words - wb < kBatchTile ? total_words - wb : kBatchTile) - 1);
        // the tile's bytes are the <= 2 KiB after its first base: 128 aligned chunks from lo16, plus chunk 128 when the
        // first base is not 16-byte aligned; clipped at the end of the buffer.  Known from the record alone.
        const uintptr_t lo = sp + rec.base0, lo16 = lo & ~(uintptr_t)15;
        const uintptr_t hi = rec.base0 + kBatchTile * 32 < seq_end ? lo + kBatchTile * 32 : sp + seq_end;
        const unsigned nchunk = hi > lo16 ? (unsigned)((hi - lo16 + 15) >> 4) : 0u; // <= 129
        const unsigned long long s = rec.owner + lane < count ? rec.owner + lane : count;
        // Only the entries the tile can need are fetched: the owner, the sequences that start inside the tile, and the first
        // one past it -- the next tile's owner is the last sequence that starts at or before word wb + 64, so that is
        // (next owner - owner) + 2 entries (14 for 150-base reads: 2 cache lines per table instead of 8).  Lanes beyond
        // that hold an entry "past everything".
        const unsigned long long span = (tile + 1 < ntiles ? recs[tile + 1].owner : count) - rec.owner + 2;
        const unsigned need = span < 64 ? (unsigned)span : 64u;
        unsigned long long wo = ~0ull, so = 0;
        if constexpr (ABL & 1) { wo = wb + 5ull * lane; so = rec.base0 + 150ull * lane; }
        else if (lane < need) { wo = word_offsets[s]; so = offsets[s]; }
        const u32x4 zero4 = {0u, 0u, 0u, 0u};
        const u32x4 v0 = lane < nchunk ? __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(lo16 + 16 * (uintptr_t)lane)) : zero4;
        const u32x4 v1 = lane + 64 < nchunk ? __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(lo16 + 16 * (uintptr_t)(lane + 64))) : zero4;
        u32x4 v2 = zero4;
        if (nchunk > 128 && lane == 0) v2 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(lo16 + 2048));
        if (rec.avail >= kBatchTile * 32 && (lo & 15) == 0) {
            // fast tile (wave-uniform): 64 full words inside one sequence = a plain 2 KiB bulk encode
            uint32_t bad = 0;
            uint32_t *o32 = reinterpret_cast<uint32_t *>(out + wb);
            const uint32_t c0 = enc16(v0, bad), c1 = enc16(v1, bad);
            store_packed(c0, o32 + lane);
            store_packed(c1, o32 + 64 + lane);
            if (__builtin_expect(residue_is_bad(bad), 0)) {
                rescan_bytes(seq, rec.base0 + 16 * lane, 16, slot);
                rescan_bytes(seq, rec.base0 + 16 * (lane + 64), 16, slot);
            }
            continue;
        }
        // code words of the chunks -> strip (chunks past the buffer end and the funnel's slack are zero)
        uint32_t b0 = 0, b1 = 0, b2 = 0;
        const u32x4 v0m = (tile == 0 && lane == 0) ? mask_lead_bytes(v0, (unsigned)(lo - lo16)) : v0; // see mask_lead_bytes
        const uint32_t c0 = enc16(v0m, b0), c1 = enc16(v1, b1), c2 = enc16(v2, b2);
        wave_lds_fence(); // the previous trip's strip readers are done
        uint32_t *mine = strip_slot(my.strip, lane); // chunk `lane`; chunk lane + 64 has the same parity: 32 dwords further
        mine[0] = c0;
        mine[32] = c1;
        if (lane == 0) *strip_slot(my.strip, 128) = c2; // (chunks past nchunk hold zeros or stale codes: stream_cut masks them off)
        if (__builtin_expect(residue_is_bad(b0) && lane < nchunk, 0)) rescan_chunk(seq, lo16, lane, seq_begin, seq_end, slot);
        if (__builtin_expect(residue_is_bad(b1) && lane + 64 < nchunk, 0)) rescan_chunk(seq, lo16, lane + 64, seq_begin, seq_end, slot);
        if (__builtin_expect(residue_is_bad(b2) && lane == 0 && nchunk > 128, 0)) rescan_chunk(seq, lo16, 128, seq_begin, seq_end, slot);
        unsigned base_rel, nb;
        if constexpr (ABL & 2) { base_rel = 30u * lane + (unsigned)(wo & 1); nb = 30u; wave_lds_fence(); }
        else tile_lookup(offsets, word_offsets, count, rec, wo, so, wb, wb + last, my.pads, base_rel, nb); // its fences order the strip stores too
        if (lane <= last) {
            const unsigned long long word = stream_cut(my.strip, (unsigned)(lo - lo16) + base_rel, nb);
            store_packed(word, out + wb + lane);
        }
    }
}

template <int ABL>
__global__ void __launch_bounds__(kBlock)
decode_batch2_kernel(const unsigned long long *__restrict__ words, const unsigned long long *__restrict__ word_offsets,
                     const unsigned long long *__restrict__ offsets, unsigned long long count,
                     unsigned long long total_words, const TileRec *__restrict__ recs, uint8_t *__restrict__ out) {
    __shared__ Batch2Lds lds[kBatchWaves];
    const unsigned lane = threadIdx.x & 63, wave = wave_in_block();
    Batch2Lds &my = lds[wave];
    const unsigned long long ntiles = (total_words + kBatchTile - 1) / kBatchTile;
    const uintptr_t op = reinterpret_cast<uintptr_t>(out);
    for (unsigned long long tile = (unsigned long long)blockIdx.x * kBatchWaves + wave; tile < ntiles;
         tile += (unsigned long long)gridDim.x * kBatchWaves) {
        const unsigned long long wb = tile * kBatchTile;
        const unsigned last = (unsigned)((total_words - wb < kBatchTile ? total_words - wb : kBatchTile) - 1);
        const unsigned long long word = __builtin_nontemporal_load(words + wb + (lane < last ? lane : last)); // independent of the lookup
        TileRec rec;
        if constexpr (ABL & 8) { rec.owner = tile * 12; rec.base0 = tile * 1900 + offsets[0]; rec.avail = 150; }
        else rec = recs[tile];
        const uintptr_t lo = op + rec.base0, lo16 = lo & ~(uintptr_t)15;
        if (rec.avail >= kBatchTile * 32 && (lo & 15) == 0) {
            // fast tile (wave-uniform): 64 full words inside one sequence; the words cross the strip once so that
            // lane l owns 16-base groups l and l+64
            wave_lds_fence();
            reinterpret_cast<unsigned long long *>(my.strip)[lane] = word;
            wave_lds_fence();
            const uint32_t h0 = my.strip[lane], h1 = my.strip[64 + lane];
            uint8_t *dst = out + rec.base0;
            store_group<true, true>(dst + 16 * lane, dec16(h0));
            store_group<true, true>(dst + 16 * (lane + 64), dec16(h1));
            continue;
        }
        const unsigned long long s = rec.owner + lane < count ? rec.owner + lane : count;
        const unsigned long long span = (tile + 1 < ntiles ? recs[tile + 1].owner : count) - rec.owner + 2; // see encode_batch2_kernel
        const unsigned need = span < 64 ? (unsigned)span : 64u;
        unsigned long long wo = ~0ull, so = 0;
        if constexpr (ABL & 1) { wo = wb + 5ull * lane; so = rec.base0 + 150ull * lane; }
        else if (lane < need) { wo = word_offsets[s]; so = offsets[s]; }
        wave_lds_fence(); // the previous trip's strip readers are done
        strip_zero(my.strip, lane);
        unsigned base_rel, nb;
        if constexpr (ABL & 2) { base_rel = 30u * lane + (unsigned)(wo & 1); nb = 30u; wave_lds_fence(); }
        else tile_lookup(offsets, word_offsets, count, rec, wo, so, wb, wb + last, my.pads, base_rel, nb); // fences: zeroing before the ORs
        const unsigned end_rel = (unsigned)__builtin_amdgcn_readlane((int)(base_rel + nb), (int)last);
        uintptr_t hi = lo + end_rel;
        uintptr_t lo_w = lo;
        if constexpr (ABL & 4) { lo_w = (lo + 15) & ~(uintptr_t)15; hi &= ~(uintptr_t)15; }
        if (lane <= last) strip_or_word(my.strip, 2u * ((unsigned)(lo - lo16) + base_rel), word, nb);
        wave_lds_fence();
        strip_drain(my.strip, my.edge, out, lo16, lo_w, hi, lane);
    }
}



// ==== decode_batch_plan_lines_kernel, round 4's line- / chunk-owning plan decode ====

// ---- line-owning plan decode ------------------------------------------------------------------------------------------------
// decode_batch_plan_kernel's tiles are 64 WORDS: a tile's output run starts and ends wherever its words do, so two neighbouring
// waves write into the same 128-byte line (and, through the two unaligned 16-byte edge stores, into the same 16-byte chunk).
// profiles/r02_plan_decode_ablation.txt: without the edge stores the kernel is 8 % faster; zeroing and OR-ing the strip cost
// nothing measurable.  Here a wave OWNS WHOLE LINES instead: its run is extended to the next line boundary past its last base with
// the first bases of the NEXT tile (whose first kExtraWords words it loads as well), and it leaves the part of its own first line
// before the boundary to the previous wave.  Every store is then a whole, aligned 16-byte chunk of a line no other wave touches --
// the store pattern of the bulk decode -- except at the two ends of the batch.
//   covered(t) := tile t+1 exists and the bases of its first kExtraWords words reach the line boundary at or after tile t's end.
// Tile t computes covered(t) from the extra words' pad bytes and covered(t-1) from its OWN first kExtraWords pad bytes: the same
// sums over the same bytes on both sides, so the two waves agree on who writes the shared line without communicating.  Where
// coverage fails (words of very few bases, or the batch's ends) the run keeps its own end and strip_drain's edge stores, as before.
// RESULT (round 4, tools/ab_plan_lines.py, profiles/r04_ab_plan_lines.txt): bit-identical output on every shape, and 4-18 % SLOWER
// (150-base reads: 0.2318 ms against 0.2069 ms).  The ablation's 8 % was the price of issuing the two edge stores, not of sharing the
// lines; what this form adds -- a second load + pad lookup per tile, a third drain iteration for the up to 137 chunks of the longer
// run, eight more strip ORs -- costs more than the shared lines did.  Evidence build only; the word-tile kernel ships.
constexpr unsigned kExtraWords = 8;                 // 8 full words = 256 bases >= 127: enough for every batch whose words average >= 16 bases
constexpr int kLineStrip = kBatchTile * 2 + 2 * (int)kExtraWords + 8; // dwords: 64 + 8 words, the <= 15-byte lead, strip_or_word's third dword

// strip_drain for a run [lo, hi) whose ends are line boundaries except at the ends of the batch: chunks of an UNSHARED line are stored
// nt like the interior; a run of up to 2 KiB + 127 bytes has up to 136 chunks (three per lane).
template <int POLICY>
__device__ __forceinline__ void strip_drain_owned(const uint32_t *strip, uint8_t (*edge)[16], uint8_t *__restrict__ out, uintptr_t lo16, uintptr_t lo, uintptr_t hi, unsigned lane) {
    const unsigned nchunk = (unsigned)((hi - lo16 + 15) >> 4); // <= 137
    const uintptr_t op = reinterpret_cast<uintptr_t>(out);
    if (hi - lo >= 16) { // wave-uniform
        const unsigned lead = (unsigned)(lo - lo16), tail = (unsigned)(hi - lo16);
        const unsigned c0 = lead ? 1u : 0u, c1 = tail >> 4; // whole chunks: c0 <= c < c1
        // chunks on a line this run shares with a neighbour: the first line when lo is not a line boundary (c < cl0), the last when hi is not (c >= cl1)
        const unsigned cl0 = (lo & 127) ? (unsigned)((((lo | 127) + 1) - lo16) >> 4) : 0u;
        const uintptr_t last_line = (hi - 1) & ~(uintptr_t)127;
        const unsigned cl1 = (hi & 127) ? (last_line > lo16 ? (unsigned)((last_line - lo16) >> 4) : 0u) : ~0u;
        uint8_t *base = out + (lo16 - op); // derived from `out`: global (not flat) stores
        if (lead == 0 && (tail & 15u) == 0) { // wave-uniform: whole chunks only -- every tile but the batch's first and last, where coverage holds
#pragma unroll
            for (int j = 0; j < 3; ++j) {
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

// GRAN = 128: whole lines as described; GRAN = 16: the same rule at 16-byte chunk granularity -- a wave completes its last partial
// CHUNK with the next tile's first bases and leaves its own leading partial chunk to its predecessor: lines are still shared, but the
// two unaligned edge stores per tile become one aligned store (one or two extra words suffice for reads).
template <int POLICY, unsigned GRAN>
__global__ void __launch_bounds__(kBlock)
decode_batch_plan_lines_kernel(const unsigned long long *__restrict__ words, const unsigned long long *__restrict__ tile_base, const uint8_t *__restrict__ P,
                               unsigned long long total_words, uint8_t *__restrict__ out) {
    __shared__ uint32_t strips[kBatchWaves][kLineStrip];
    __shared__ __attribute__((aligned(16))) uint8_t edge[kBatchWaves][2][16];
    const unsigned lane = threadIdx.x & 63, wave = wave_in_block();
    uint32_t *strip = strips[wave];
    const unsigned long long ntiles = (total_words + kBatchTile - 1) / kBatchTile;
    const uintptr_t op = reinterpret_cast<uintptr_t>(out);
    for (unsigned long long tile = (unsigned long long)blockIdx.x * kBatchWaves + wave; tile < ntiles; tile += (unsigned long long)gridDim.x * kBatchWaves) {
        const unsigned long long wb = tile * kBatchTile;
        const unsigned last = (unsigned)((total_words - wb < kBatchTile ? total_words - wb : kBatchTile) - 1);
        const unsigned long long word = __builtin_nontemporal_load(words + wb + (lane < last ? lane : last));
        const uint32_t n = lane <= last ? (uint32_t)P[wb + lane + 1] : 0u;
        // the next tile's first words (every lane loads one of them: eight addresses, one request)
        const unsigned long long xw = wb + kBatchTile + (lane & (kExtraWords - 1));
        const bool has_x = xw < total_words;
        const unsigned long long xword = words[has_x ? xw : total_words - 1]; // plain load: the next tile's wave reads the same line
        const uint32_t xnb = has_x && lane < kExtraWords ? 32u - (uint32_t)P[xw + 1] : 0u;
        const unsigned long long base0 = tile_base[tile];
        const bool dense = __ballot(n != 0u) == 0ull;
        const uintptr_t lo = op + base0;
        if (dense && last == 63u && (lo & (GRAN - 1)) == 0) {
            // fast tile (wave-uniform): 2 KiB of whole lines, exactly its own words' bases (both neighbours see a boundary here)
            wave_lds_fence();
            reinterpret_cast<unsigned long long *>(strip)[lane] = word;
            wave_lds_fence();
            const uint32_t h0 = strip[lane], h1 = strip[64 + lane];
            uint8_t *dst = out + base0;
            store_group<true, true>(dst + 16 * lane, dec16(h0));
            store_group<true, true>(dst + 16 * (lane + 64), dec16(h1));
            continue;
        }
        // one scan for both running sums: pad bytes of the tile's words (low half) and bases of the extra words (high half)
        const uint32_t incl = wave_inclusive_sum(n | (xnb << 16));
        const unsigned nb = 32u - n, base_rel = 32u * lane - ((incl & 0xFFFFu) - n);
        const unsigned end_rel = (unsigned)__builtin_amdgcn_readlane((int)(base_rel + nb), (int)last);
        const unsigned first8 = (unsigned)__builtin_amdgcn_readlane((int)(base_rel + nb), (int)(last < kExtraWords - 1 ? last : kExtraWords - 1)); // bases of this tile's first words
        const unsigned next8 = (unsigned)__builtin_amdgcn_readlane((int)(incl >> 16), 63);                                                     // bases of the next tile's first words
        const uintptr_t hi = lo + end_rel;
        const unsigned gap_lo = (unsigned)(0 - lo) & (GRAN - 1), gap_hi = (unsigned)(0 - hi) & (GRAN - 1);
        const bool prev_covers = tile > 0 && first8 >= gap_lo;               // covered(tile - 1), from this tile's own pad bytes
        const bool covers = wb + kBatchTile < total_words && next8 >= gap_hi; // covered(tile), from the extra words' pad bytes
        const uintptr_t own_lo = prev_covers ? lo + gap_lo : lo, own_hi = covers ? hi + gap_hi : hi;
        if (own_hi <= own_lo) continue; // (wave-uniform) a last tile that lies inside the line its predecessor completed
        const uintptr_t lo16 = lo & ~(uintptr_t)15;
        wave_lds_fence(); // the previous tile's strip readers are done
#pragma unroll
        for (int j = 0; j < (kLineStrip + 63) / 64; ++j)
            if (lane + 64 * j < (unsigned)kLineStrip) strip[lane + 64 * j] = 0u;
        wave_lds_fence();
        const unsigned lead_bits = 2u * (unsigned)(lo - lo16);
        if (lane <= last) strip_or_word(strip, lead_bits + 2u * base_rel, word, nb);
        if (covers && xnb) strip_or_word(strip, lead_bits + 2u * (end_rel + ((incl >> 16) - xnb)), xword, xnb);
        wave_lds_fence();
        const uintptr_t own16 = own_lo & ~(uintptr_t)15;
        strip_drain_owned<POLICY>(strip + ((own16 - lo16) >> 4), edge[wave], out, own16, own_lo, own_hi, lane);
    }
}

