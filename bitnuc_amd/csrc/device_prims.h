// device_prims.h -- gfx950 device primitives shared by every kernel family of libbitnuc_hip.so:
// the per-dword 2-bit pack / unpack (v_perm LUT + v_dot4 compaction), the validity residue and the
// first-invalid-byte latch (packing/avx.rs:86-91), cache-policy load / store helpers, the wave-private
// LDS fence, the XCD-aware tile order and the single-launch reduction protocol.  No kernels here: a
// kernel header is included by exactly one translation unit (codec.hip, kmer.hip, batch.hip, analysis.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bitnuc_dev {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x4 u32x4_u __attribute__((aligned(1))); // any byte address (gfx950 unaligned-access mode)
typedef uint32_t u32_u __attribute__((aligned(1)));
typedef u32x2 u32x2_u __attribute__((aligned(1)));

constexpr unsigned long long kNoBad = ~0ull;
constexpr int kBlock = 256;

// ---------------------------------------------------------------------------------
// per-dword primitives
// ---------------------------------------------------------------------------------

// 4 ASCII bases -> one byte of 4 codes (base 0 in bits 0-1).  `bad` accumulates the
// validity residue: (bad & 0xFCFCFCFC) != 0 <=> some byte was not in ACGTacgt.
__device__ __forceinline__ uint32_t enc4(uint32_t x, uint32_t &bad) {
    const uint32_t sel = x & 0x07070707u;
    // LUT[idx] : 1->'A'(0x40|0) 3->'C'(0x40|1) 7->'G'(0x40|2) 4->'T'(0x50|3); others 0x04 (bit 2 = invalid)
    const uint32_t t = __builtin_amdgcn_perm(0x42040453u, 0x41044004u, sel);
    const uint32_t d = t ^ (x & 0xD8D8D8D8u);
    bad |= d;
    return __builtin_amdgcn_udot4(d, 0x40100401u, 0u, false);
}

__device__ __forceinline__ bool residue_is_bad(uint32_t bad) { return (bad & 0xFCFCFCFCu) != 0u; }

__device__ __forceinline__ uint32_t enc16(u32x4 v, uint32_t &bad) {
    const uint32_t r0 = enc4(v.x, bad), r1 = enc4(v.y, bad), r2 = enc4(v.z, bad), r3 = enc4(v.w, bad);
    return r0 | (r1 << 8) | (r2 << 16) | (r3 << 24);
}

__device__ __forceinline__ bool valid_base(uint32_t b) {
    const uint32_t u = b & 0xDFu;
    return u == 'A' || u == 'C' || u == 'G' || u == 'T';
}

__device__ __forceinline__ uint32_t code_of(uint32_t b) { return ((b >> 1) ^ (b >> 2)) & 3u; }

// slow path (rare): re-read `nbytes` bytes starting at absolute index `start` and latch
// the first invalid one.  One non-unrolled copy per kernel keeps the hot loop's
// register/SGPR footprint small.
// The slot holds (absolute byte index << 8) | offending byte: atomicMin orders by index, and the
// byte travels with it, so reporting never has to re-read an input the caller may have reused.
__device__ __forceinline__ void latch_bad(unsigned long long *slot, unsigned long long index, uint32_t byte) {
    atomicMin(slot, (index << 8) | (byte & 0xFFu));
}

__device__ __forceinline__ void rescan_bytes(const uint8_t *seq, unsigned long long start, unsigned nbytes,
                                             unsigned long long *slot) {
#pragma unroll 1
    for (unsigned i = 0; i < nbytes; ++i) {
        const uint32_t b = seq[start + i];
        if (!valid_base(b)) {
            latch_bad(slot, start + i, b);
            return;
        }
    }
}

// threadIdx.x >> 6 is the same in all lanes of a wave; reading it through readfirstlane tells the
// compiler, so everything derived from it (tile index, tile base address, record loads) is
// computed once on the scalar unit instead of per lane in 64-bit VALU arithmetic.
__device__ __forceinline__ unsigned wave_in_block() { return (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Single-launch grid reductions: every workgroup adds its partial to context-owned accumulators
// with device-scope atomics (performed at the coherence point, so they need no cache fences --
// an agent-scope __threadfence() per workgroup costs an L2 write-back and made these kernels 5x
// slower), then one thread draws a ticket; whoever draws the last one swaps the accumulators
// back to zero (ready for the next launch on the stream) and writes the result.  No memset
// before, no finishing launch after.
// Call draw_last_ticket from ONE thread, after a __syncthreads() that follows the workgroup's
// add_performed() calls.
// Ordering argument (why relaxed atomics suffice, no fence): every access to the accumulators and the ticket is a
// device-scope atomic RMW, performed at the L2 / memory side in arrival order, never a cached load or store.  add_performed
// uses the RETURNING form and consumes the returned value, so the wave cannot reach the barrier before its add has been
// performed at the coherence point; the ticket RMW is issued after that barrier, hence after every add of its workgroup.
// The workgroup that draws the last ticket therefore reads (atomicExch) accumulators that already hold every workgroup's
// contribution.  Nothing else is communicated between workgroups (no plain data), which is what a fence would be for.
template <class T> __device__ __forceinline__ void add_performed(T *acc, T v) {
    const T prev = atomicAdd(acc, v); // returning form: the value can only arrive once the add has been performed
    // consuming the returned value forces the wait for it; the "memory" clobber keeps the compiler from moving any other memory
    // access (the ticket RMW in particular) across this point
    if constexpr (sizeof(T) == 8) asm volatile("" ::"v"((uint32_t)prev), "v"((uint32_t)(prev >> 32)) : "memory");
    else asm volatile("" ::"v"(prev) : "memory");
}
__device__ __forceinline__ bool draw_last_ticket(unsigned *ticket) {
    if (atomicAdd(ticket, 1u) != gridDim.x - 1) return false;
    atomicExch(ticket, 0u);
    return true;
}

// 8 bits (4 codes) -> 4 ASCII bytes.
__device__ __forceinline__ uint32_t dec4(uint32_t v) {
    uint32_t s = (v | (v << 12)) & 0x000F000Fu;
    s = (s | (s << 6)) & 0x03030303u;
    return __builtin_amdgcn_perm(0u, 0x54474341u /* "ACGT" */, s);
}

__device__ __forceinline__ u32x4 dec16(uint32_t w) {
    u32x4 o;
    o.x = dec4(w & 0xFFu);
    o.y = dec4(__builtin_amdgcn_ubfe(w, 8, 8));
    o.z = dec4(__builtin_amdgcn_ubfe(w, 16, 8));
    o.w = dec4(w >> 24);
    return o;
}

// ---------------------------------------------------------------------------------
// memory helpers
// ---------------------------------------------------------------------------------
template <bool NT, bool ALIGNED>
__device__ __forceinline__ u32x4 load_group(const uint8_t *p) {
    if constexpr (ALIGNED) {
        if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
        else return *reinterpret_cast<const u32x4 *>(p);
    } else {
        return *reinterpret_cast<const u32x4_u *>(p);
    }
}

template <bool NT, bool ALIGNED>
__device__ __forceinline__ void store_group(uint8_t *p, u32x4 v) {
    if constexpr (ALIGNED) {
        if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
        else *reinterpret_cast<u32x4 *>(p) = v;
    } else {
        *reinterpret_cast<u32x4_u *>(p) = v;
    }
}

template <bool NT>
__device__ __forceinline__ void store_u32(uint32_t *p, uint32_t v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

template <bool NT>
__device__ __forceinline__ uint32_t load_u32(const uint32_t *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

__device__ __forceinline__ void wave_lds_fence() {
    // LDS ops of one wave complete in order; this only stops the compiler from
    // moving the ds_read above the ds_write of the other lanes.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------
// tile -> workgroup mapping
// ---------------------------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one).  With XCD the
// first pass gives each XCD one contiguous eighth of the tiles instead of every eighth
// tile (speed only; any placement is correct).  Bijective for any grid size.
template <bool XCD>
__device__ __forceinline__ unsigned long long first_tile(unsigned b, unsigned grid) {
    if constexpr (!XCD) return b;
    const unsigned q = grid >> 3, r = grid & 7, x = b & 7;
    return (unsigned long long)(x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

} // namespace bitnuc_dev
