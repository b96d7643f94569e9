// util.hip — small device helpers of the host boundary (no rasterizer arithmetic).
#include "gdr_common.h"

namespace gdr {
namespace {

// flag |= 1 if any 32-bit word of a differs from b (bitwise: -0.0 != 0.0, NaN payloads compared as bits)
__global__ __launch_bounds__(GDR_BLOCK) void words_differ_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b,
                                                                  uint64_t n16, const uint32_t* __restrict__ a_tail,
                                                                  const uint32_t* __restrict__ b_tail, uint32_t n_tail,
                                                                  uint32_t* __restrict__ flag) {
    bool diff = false;
    for (uint64_t i = (uint64_t)blockIdx.x * GDR_BLOCK + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * GDR_BLOCK) {
        const uint4 x = a[i], y = b[i];
        diff |= (x.x != y.x) | (x.y != y.y) | (x.z != y.z) | (x.w != y.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < n_tail) diff |= a_tail[threadIdx.x] != b_tail[threadIdx.x];
    if (__ballot(diff) != 0ull && (threadIdx.x & 63u) == 0u) atomicOr(flag, 1u);
}

}  // namespace

hipError_t launch_words_differ(const void* a, const void* b, uint64_t n_bytes, uint32_t* flag, hipStream_t st) {
    const uint64_t n16 = n_bytes / 16;
    const uint32_t n_tail = (uint32_t)((n_bytes % 16) / 4);
    const int blocks = (int)(n16 / GDR_BLOCK + 1 < 2048 ? n16 / GDR_BLOCK + 1 : 2048);
    hipLaunchKernelGGL(words_differ_kernel, dim3(blocks), dim3(GDR_BLOCK), 0, st, (const uint4*)a, (const uint4*)b, n16,
                       (const uint32_t*)a + 4 * n16, (const uint32_t*)b + 4 * n16, n_tail, flag);
    return hipGetLastError();
}

}  // namespace gdr
