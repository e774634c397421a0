// Experiment: does any gfx950 store cache-policy combination (sc0 / sc1 / nt) lift the write-only
// streaming rate above what plain and __builtin_nontemporal_store reach?  (decode is bound by it.)
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/store_policy tools/exp/store_policy.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int POLICY>
__device__ __forceinline__ void store16(u32x4 *p, u32x4 v) {
    if constexpr (POLICY == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

template <int POLICY, int UNROLL>
__global__ void __launch_bounds__(256) fill(u32x4 *dst, unsigned long long n16) {
    const unsigned long long tile = (unsigned long long)256 * UNROLL;
    const u32x4 v = {0x41414141u, 0x43434343u, 0x47474747u, 0x54545454u};
    for (unsigned long long t = blockIdx.x; t < n16 / tile; t += gridDim.x)
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) store16<POLICY>(dst + t * tile + u * 256 + threadIdx.x, v);
}

template <int POLICY>
double run(u32x4 *a, u32x4 *b, unsigned long long n16) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<float> ms;
    const unsigned grid = (unsigned)(n16 / (256 * 2));
    for (int i = 0; i < 12; ++i) {
        hipEventRecord(e0);
        fill<POLICY, 2><<<grid, 256>>>(i & 1 ? a : b, n16);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        if (i >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main() {
    const unsigned long long bytes = 1ull << 30, n16 = bytes / 16;
    u32x4 *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    const char *names[8] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
    double ms[8];
    for (int rep = 0; rep < 2; ++rep) {
        ms[0] = run<0>(a, b, n16); ms[1] = run<1>(a, b, n16); ms[2] = run<2>(a, b, n16); ms[3] = run<3>(a, b, n16);
        ms[4] = run<4>(a, b, n16); ms[5] = run<5>(a, b, n16); ms[6] = run<6>(a, b, n16); ms[7] = run<7>(a, b, n16);
        for (int p = 0; p < 8; ++p) printf("%-11s %.4f ms  %.0f GB/s\n", names[p], ms[p], bytes / ms[p] / 1e6);
        printf("--\n");
    }
    return 0;
}
