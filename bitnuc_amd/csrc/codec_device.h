// codec_device.h -- gfx950 device code for bulk 2-bit encode / decode.
//
// What is computed (bit-exact target):
//   encode : word[w] = sum_i code(seq[32w+i]) << 2i, code A/a,C/c,G/g,T/t -> 0..3
//            (reference: src/utils/packing/naive.rs:8-18, avx.rs:130-151)
//   decode : out[32w+i] = "ACGT"[(word[w] >> 2i) & 3]
//            (reference: src/utils/unpacking/naive.rs:12-22, avx.rs:116-153)
//
// How (MI355X-first, nothing here is a translation of the reference's SIMD):
//   * the unit of work is a 16-base "group": 16 input bytes <-> one u32 (half a
//     u64 word, little-endian), so a lane's global access is one dwordx4 on the
//     ASCII side and adjacent lanes touch adjacent 16-byte groups: every
//     wave-instruction is a fully coalesced 1 KiB (ASCII) / 256 B (packed) span;
//   * per dword of 4 bases: index = b & 7 (A=1,C=3,T=4,G=7, case-insensitive)
//     -> v_perm_b32 as an 8-entry byte LUT yields code | (ascii & 0xD8); XOR with
//     (x & 0xD8D8D8D8) cancels the ASCII bits iff the byte is a valid base, so the
//     same word carries the code (bits 0-1) and the validity residue (bits 2-7);
//     v_dot4_u32_u8 with weights {1,4,16,64} compacts 4 codes into one byte;
//   * an invalid byte is rare: lanes that see a residue take a slow path that
//     atomicMin's (absolute byte index << 8 | byte) into a per-launch slot, which gives the
//     reference's "first invalid byte in sequence order" (avx.rs:86-91) without
//     serialising the stream;
//   * optional wave-private LDS transpose (XPOSE) turns 4 coalesced 4-byte
//     accesses per lane on the packed side into one 16-byte access;
//   * strided wide accesses (a lane owning 2 or 4 adjacent groups) were measured and
//     dropped: 64 B/lane loads ran encode at 4.3 TB/s, 32-64 B/lane stores ran decode
//     at 1.5-3.0 TB/s, against 6.2 / 6.5 TB/s for the coalesced forms (profiles/).
#pragma once
#include "device_prims.h"

namespace bitnuc_dev {

// What the tiled loops leave over, done by ONE workgroup: the 16-base groups from `first_group` on (bounds-checked, plain
// path), the last 1..15 bases, and the zero upper half of the final u64 when the number of 16-base groups is odd.
template <int BLOCK, bool ALIGNED>
__device__ __forceinline__ void encode_tail(const uint8_t *__restrict__ seq, uint32_t *__restrict__ out32, unsigned long long len,
                                            unsigned long long first_group, unsigned long long *__restrict__ slot) {
    const unsigned t = threadIdx.x;
    const unsigned long long n16 = len >> 4;
    for (unsigned long long g = first_group + t; g < n16; g += BLOCK) {
        const u32x4 v = load_group<false, ALIGNED>(seq + (g << 4));
        uint32_t b = 0;
        const uint32_t r = enc16(v, b);
        if (residue_is_bad(b)) rescan_bytes(seq, g << 4, 16, slot);
        out32[g] = r;
    }
    if (t == 0) {
        const unsigned rem = (unsigned)(len & 15);
        unsigned long long ngroups = n16;
        if (rem) {
            uint32_t r = 0;
            bool flagged = false;
            for (unsigned i = 0; i < rem; ++i) {
                const uint32_t b = seq[(n16 << 4) + i];
                if (!valid_base(b) && !flagged) {
                    latch_bad(slot, (n16 << 4) + i, b);
                    flagged = true;
                }
                r |= code_of(b) << (2 * i);
            }
            out32[n16] = r;
            ngroups = n16 + 1;
        }
        if (ngroups & 1) out32[ngroups] = 0;
    }
}

// ---------------------------------------------------------------------------------
// encode
// ---------------------------------------------------------------------------------
// Geometry: a tile is BLOCK*UNROLL groups.  Thread t, round u owns group
// tile + u*BLOCK + t, so one round of a wave is one contiguous 1 KiB span of ASCII
// (dwordx4 per lane) and one 256 B span of packed output (dword per lane).
// XPOSE (UNROLL==4): the wave's 4 results per lane cross a wave-private LDS strip so
// each lane stores 16 contiguous packed bytes (one dwordx4) instead of 4 dwords.
template <int UNROLL, int BLOCK, bool NTLD, bool NTST, bool ALIGNED, bool XPOSE, bool XCD>
__global__ void __launch_bounds__(BLOCK)
encode_kernel(const uint8_t *__restrict__ seq, uint32_t *__restrict__ out32, unsigned long long len,
              unsigned long long *__restrict__ slot) {
    static_assert(!XPOSE || UNROLL == 4, "XPOSE needs UNROLL=4");
    constexpr unsigned long long TILE = (unsigned long long)BLOCK * UNROLL;
    const unsigned long long n16 = len >> 4;
    const unsigned long long full_tiles = n16 / TILE;
    const unsigned t = threadIdx.x;
    __shared__ uint32_t strip[XPOSE ? BLOCK * 4 : 1];

    for (unsigned long long tile = first_tile<XCD>(blockIdx.x, gridDim.x); tile < full_tiles; tile += gridDim.x) {
        const unsigned long long g0 = tile * TILE;
        if constexpr (XPOSE) {
            const unsigned wave = t >> 6, lane = t & 63;
            const unsigned long long gw = g0 + (unsigned long long)wave * 256;
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = load_group<NTLD, true>(seq + ((gw + u * 64 + lane) << 4));
            uint32_t bad = 0;
            uint32_t *ws = strip + wave * 256;
#pragma unroll
            for (int u = 0; u < 4; ++u) ws[u * 64 + lane] = enc16(v[u], bad);
            if (__builtin_expect(residue_is_bad(bad), 0)) {
#pragma unroll 1
                for (int u = 0; u < 4; ++u) rescan_bytes(seq, (gw + u * 64 + lane) << 4, 16, slot);
            }
            wave_lds_fence();
            const u32x4 r = *reinterpret_cast<const u32x4 *>(ws + 4 * lane);
            wave_lds_fence();
            store_group<NTST, true>(reinterpret_cast<uint8_t *>(out32 + gw + 4 * lane), r);
        } else {
            u32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                v[u] = load_group<NTLD, ALIGNED>(seq + ((g0 + (unsigned long long)u * BLOCK + t) << 4));
            uint32_t bad = 0;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                store_u32<NTST>(out32 + g0 + (unsigned long long)u * BLOCK + t, enc16(v[u], bad));
            if (__builtin_expect(residue_is_bad(bad), 0)) {
#pragma unroll 1
                for (int u = 0; u < UNROLL; ++u) rescan_bytes(seq, (g0 + (unsigned long long)u * BLOCK + t) << 4, 16, slot);
            }
        }
    }

    // groups past the last full tile: one block, bounds-checked, plain path
    if (blockIdx.x == (unsigned)(full_tiles % gridDim.x)) encode_tail<BLOCK, ALIGNED>(seq, out32, len, full_tiles * TILE, slot);
}

// (encode_quad_kernel -- 16-byte stores by a register quad transpose, round 3 -- and encode_ballot_kernel -- the lane-per-base + ballot
// formulation north_star sketches, 13.5x slower -- : csrc/evidence/codec_evidence.h, evidence build only)

// ---------------------------------------------------------------------------------
// decode
// ---------------------------------------------------------------------------------
// Same geometry seen from the packed side: thread t, round u loads one u32 half-word
// and writes one 16-byte ASCII group (dwordx4), both fully coalesced.  XPOSE: dwordx4
// load of 4 consecutive half-words per lane, transposed through a wave-private LDS
// strip so that each of the 4 ASCII stores of a wave is still one contiguous 1 KiB span.
template <int UNROLL, int BLOCK, bool NTLD, bool NTST, bool ALIGNED, bool XPOSE, bool XCD>
__global__ void __launch_bounds__(BLOCK)
decode_kernel(const uint32_t *__restrict__ in32, uint8_t *__restrict__ out, unsigned long long n_bases) {
    static_assert(!XPOSE || UNROLL == 4, "XPOSE needs UNROLL=4");
    constexpr unsigned long long TILE = (unsigned long long)BLOCK * UNROLL;
    const unsigned long long n16 = n_bases >> 4;
    const unsigned long long full_tiles = n16 / TILE;
    const unsigned t = threadIdx.x;
    __shared__ uint32_t strip[XPOSE ? BLOCK * 4 : 1];

    for (unsigned long long tile = first_tile<XCD>(blockIdx.x, gridDim.x); tile < full_tiles; tile += gridDim.x) {
        const unsigned long long g0 = tile * TILE;
        if constexpr (XPOSE) {
            const unsigned wave = t >> 6, lane = t & 63;
            const unsigned long long gw = g0 + (unsigned long long)wave * 256;
            uint32_t *ws = strip + wave * 256;
            const u32x4 w = load_group<NTLD, true>(reinterpret_cast<const uint8_t *>(in32 + gw + 4 * lane));
            *reinterpret_cast<u32x4 *>(ws + 4 * lane) = w;
            wave_lds_fence();
            uint32_t h[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) h[u] = ws[u * 64 + lane];
            wave_lds_fence();
#pragma unroll
            for (int u = 0; u < 4; ++u) store_group<NTST, ALIGNED>(out + ((gw + u * 64 + lane) << 4), dec16(h[u]));
        } else {
            uint32_t h[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) h[u] = load_u32<NTLD>(in32 + g0 + (unsigned long long)u * BLOCK + t);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                store_group<NTST, ALIGNED>(out + ((g0 + (unsigned long long)u * BLOCK + t) << 4), dec16(h[u]));
        }
    }

    if (blockIdx.x == (unsigned)(full_tiles % gridDim.x)) {
        for (unsigned long long g = full_tiles * TILE + t; g < n16; g += BLOCK)
            store_group<false, ALIGNED>(out + (g << 4), dec16(in32[g]));
        if (t == 0) {
            const unsigned rem = (unsigned)(n_bases & 15);
            if (rem) {
                const uint32_t w = in32[n16];
                for (unsigned i = 0; i < rem; ++i) out[(n16 << 4) + i] = "ACGT"[(w >> (2 * i)) & 3];
            }
        }
    }
}

// (decode_x2_kernel -- 8-byte loads + an LDS transpose, round 2: csrc/evidence/codec_evidence.h, evidence build only)

// ---------------------------------------------------------------------------------
// synthetic input generator (seeded, counter-based; regenerable on the host)
// ---------------------------------------------------------------------------------
__host__ __device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ unsigned long long gen_word(unsigned long long seed, unsigned long long widx, int flags) {
    if (flags & 1) return 0xE4E4E4E4E4E4E4E4ull; // "ACGT" cyclic (bases[i % 4])
    return mix64(seed + (widx + 1) * 0x9E3779B97F4A7C15ull);
}

// flags bit 1: lower-case mix (SURVEY 8d parity variant, p = 0.25): base i is lower case iff the 2-bit field i of a second,
// independent word stream (seed ^ kCaseSalt) is zero.  Returns 0x20 in the bytes of a 16-base group that are lower case.
constexpr unsigned long long kCaseSalt = 0xC0FFEE5EEDC0DE55ull;
__device__ __forceinline__ u32x4 case_bits16(uint32_t bits) { // 32 bits = 16 two-bit fields -> 16 bytes of 0x20 / 0x00
    // field == 0  <=>  neither of its two bits is set
    const uint32_t z = ~(bits | (bits >> 1)) & 0x55555555u; // bit 2j set iff field j is zero
    u32x4 o;
    const uint32_t parts[4] = {z & 0xFFu, (z >> 8) & 0xFFu, (z >> 16) & 0xFFu, z >> 24};
    uint32_t r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t sft = (parts[q] | (parts[q] << 12)) & 0x000F000Fu; // the dec4 spread: field j of the byte -> byte j
        sft = (sft | (sft << 6)) & 0x03030303u;
        r[q] = (sft & 0x01010101u) << 5;
    }
    o.x = r[0]; o.y = r[1]; o.z = r[2]; o.w = r[3];
    return o;
}

__global__ void __launch_bounds__(kBlock)
nucgen_kernel(uint8_t *__restrict__ out, unsigned long long len, unsigned long long seed,
              unsigned long long first, int flags) {
    const unsigned long long ngroups = (len + 15) >> 4;
    for (unsigned long long g = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; g < ngroups;
         g += (unsigned long long)gridDim.x * kBlock) {
        const unsigned long long a = first + (g << 4); // absolute index of this group's first base
        const unsigned long long widx = a >> 5;
        const unsigned s = 2 * (unsigned)(a & 31);
        const unsigned long long w0 = gen_word(seed, widx, flags);
        unsigned long long bits = w0 >> s;
        if (s > 32) bits |= gen_word(seed, widx + 1, flags) << (64 - s);
        u32x4 v = dec16((uint32_t)bits);
        if (flags & 2) {
            const unsigned long long c0 = gen_word(seed ^ kCaseSalt, widx, 0);
            unsigned long long cb = c0 >> s;
            if (s > 32) cb |= gen_word(seed ^ kCaseSalt, widx + 1, 0) << (64 - s);
            v |= case_bits16((uint32_t)cb);
        }
        const unsigned long long o = g << 4;
        if (o + 16 <= len) {
            store_group<false, false>(out + o, v);
        } else {
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            for (unsigned i = 0; o + i < len; ++i) out[o + i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
        }
    }
}

// ---------------------------------------------------------------------------------
// streaming probes: the box's own HBM ceiling, measured next to the codec
// ---------------------------------------------------------------------------------
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(kBlock)
probe_read_kernel(const u32x4 *__restrict__ src, unsigned long long n16, uint32_t *__restrict__ sink) {
    constexpr unsigned long long TILE = (unsigned long long)kBlock * UNROLL;
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long full_tiles = n16 / TILE;
    for (unsigned long long tile = blockIdx.x; tile < full_tiles; tile += gridDim.x) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const u32x4 *p = src + tile * TILE + u * kBlock + threadIdx.x;
            if constexpr (NT) v[u] = __builtin_nontemporal_load(p); else v[u] = *p;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
    }
    const uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x9E3779B9u) sink[0] = r; // never true for real data; keeps the loads alive
}

template <int UNROLL, bool NT, bool NTST = NT>
__global__ void __launch_bounds__(kBlock)
probe_copy_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, unsigned long long n16) {
    constexpr unsigned long long TILE = (unsigned long long)kBlock * UNROLL;
    const unsigned long long full_tiles = n16 / TILE;
    for (unsigned long long tile = blockIdx.x; tile < full_tiles; tile += gridDim.x) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const u32x4 *p = src + tile * TILE + u * kBlock + threadIdx.x;
            if constexpr (NT) v[u] = __builtin_nontemporal_load(p); else v[u] = *p;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            u32x4 *q = dst + tile * TILE + u * kBlock + threadIdx.x;
            if constexpr (NTST) __builtin_nontemporal_store(v[u], q); else *q = v[u];
        }
    }
}

template <int UNROLL, bool NT>
__global__ void __launch_bounds__(kBlock)
probe_fill_kernel(u32x4 *__restrict__ dst, unsigned long long n16) {
    constexpr unsigned long long TILE = (unsigned long long)kBlock * UNROLL;
    const unsigned long long full_tiles = n16 / TILE;
    const u32x4 v = {0x41414141u, 0x43434343u, 0x47474747u, 0x54545454u};
    for (unsigned long long tile = blockIdx.x; tile < full_tiles; tile += gridDim.x) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            u32x4 *q = dst + tile * TILE + u * kBlock + threadIdx.x;
            if constexpr (NT) __builtin_nontemporal_store(v, q); else *q = v;
        }
    }
}


// The codec kernels' exact access shapes with no arithmetic between load and store (bench.py's *_shape probes):
// ENC: 16 B load + 4 B store per lane and round (1 : 0.25), DEC: 4 B load + 16 B store.  Same tile walk, block size,
// rounds in flight and cache policy as encode_kernel / decode_kernel.
template <int UNROLL, int BLOCK, bool NTLD, bool NTST, bool XCD = false>
__global__ void __launch_bounds__(BLOCK)
probe_enc_shape_kernel(const u32x4 *__restrict__ src, uint32_t *__restrict__ dst, unsigned long long n16) {
    constexpr unsigned long long TILE = (unsigned long long)BLOCK * UNROLL;
    const unsigned long long full_tiles = n16 / TILE;
    for (unsigned long long tile = first_tile<XCD>(blockIdx.x, gridDim.x); tile < full_tiles; tile += gridDim.x) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const u32x4 *p = src + tile * TILE + u * BLOCK + threadIdx.x;
            if constexpr (NTLD) v[u] = __builtin_nontemporal_load(p); else v[u] = *p;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) store_u32<NTST>(dst + tile * TILE + u * BLOCK + threadIdx.x, v[u].x ^ v[u].y ^ v[u].z ^ v[u].w);
    }
}

template <int UNROLL, int BLOCK, bool NTLD, bool NTST>
__global__ void __launch_bounds__(BLOCK)
probe_dec_shape_kernel(const uint32_t *__restrict__ src, u32x4 *__restrict__ dst, unsigned long long n16) {
    constexpr unsigned long long TILE = (unsigned long long)BLOCK * UNROLL;
    const unsigned long long full_tiles = n16 / TILE;
    for (unsigned long long tile = blockIdx.x; tile < full_tiles; tile += gridDim.x) {
        uint32_t h[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) h[u] = load_u32<NTLD>(src + tile * TILE + u * BLOCK + threadIdx.x);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const u32x4 v = {h[u], h[u] + 1u, h[u] + 2u, h[u] + 3u};
            u32x4 *q = dst + tile * TILE + u * BLOCK + threadIdx.x;
            if constexpr (NTST) __builtin_nontemporal_store(v, q); else *q = v;
        }
    }
}

// (probe_win_shape_kernel, the every-window kernel's no-arithmetic shape: csrc/evidence/codec_evidence.h, evidence build only)

#ifdef BITNUC_SWEEP_VARIANTS
#include "evidence/codec_evidence.h" // the formulations that lost their A/B: evidence build only
#endif

} // namespace bitnuc_dev
