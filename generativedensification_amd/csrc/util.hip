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

struct DifferPairs { int n; const uint32_t* a[GDR_DIFFER_MAX]; const uint32_t* b[GDR_DIFFER_MAX]; uint64_t words[GDR_DIFFER_MAX]; };

// the same for up to GDR_DIFFER_MAX buffer pairs in one launch (blockIdx.y = pair)
__global__ __launch_bounds__(GDR_BLOCK) void words_differ_multi_kernel(const DifferPairs p, uint32_t* __restrict__ flag) {
    const uint32_t* __restrict__ a = p.a[blockIdx.y];
    const uint32_t* __restrict__ b = p.b[blockIdx.y];
    const uint64_t n = p.words[blockIdx.y], n16 = n / 4;
    const uint4* __restrict__ a4 = reinterpret_cast<const uint4*>(a);
    const uint4* __restrict__ b4 = reinterpret_cast<const uint4*>(b);
    bool diff = false;
    for (uint64_t i = (uint64_t)blockIdx.x * GDR_BLOCK + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * GDR_BLOCK) {
        const uint4 x = a4[i], y = b4[i];
        diff |= (x.x != y.x) | (x.y != y.y) | (x.z != y.z) | (x.w != y.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (uint32_t)(n % 4)) diff |= a[4 * n16 + threadIdx.x] != b[4 * n16 + threadIdx.x];
    if (__ballot(diff) != 0ull && (threadIdx.x & 63u) == 0u) atomicOr(flag, 1u);
}

// the device tensors of one settings record as words: bg (3), viewmatrix (16), projmatrix (16), campos (3)
struct SettingsWords { const uint32_t* bg; const uint32_t* view; const uint32_t* proj; const uint32_t* campos; };
struct MatchArgs { SettingsWords cur; SettingsWords cand[GDR_REUSE_MAX]; uint64_t eligible; };

__device__ __forceinline__ const uint32_t* settings_word(const SettingsWords& w, uint32_t t) {
    return t < 3u ? w.bg + t : t < 19u ? w.view + (t - 3u) : t < 35u ? w.proj + (t - 19u) : w.campos + (t - 35u);
}

// one wave per candidate; lane t < 38 compares word t (scalar loads: the tensors may sit at any 4-byte alignment —
// `batch['bg_color'][i, j]` is a 12-byte slice of a larger tensor)
__global__ __launch_bounds__(64) void settings_match_kernel(const MatchArgs a, uint32_t* __restrict__ out) {
    const uint32_t c = blockIdx.x, t = threadIdx.x;
    bool diff = false;
    if ((a.eligible >> c) & 1ull) {
        if (t < 38u) diff = *settings_word(a.cur, t) != *settings_word(a.cand[c], t);
    } else {
        diff = true;
    }
    const bool any = __ballot(diff) != 0ull;
    if (t == 0u) out[c] = any ? 0u : 1u;
}

}  // namespace

hipError_t launch_settings_match(const gdr_settings* cur, int n, const gdr_settings* cand, uint64_t eligible, uint32_t* out,
                                 hipStream_t st) {
    if (n <= 0) return hipSuccess;
    MatchArgs a;
    auto words = [](const gdr_settings& s) {
        return SettingsWords{(const uint32_t*)s.bg, (const uint32_t*)s.viewmatrix, (const uint32_t*)s.projmatrix,
                             (const uint32_t*)s.campos};
    };
    a.cur = words(*cur);
    for (int c = 0; c < GDR_REUSE_MAX; ++c) a.cand[c] = c < n ? words(cand[c]) : a.cur;
    a.eligible = eligible;
    hipLaunchKernelGGL(settings_match_kernel, dim3(n), dim3(64), 0, st, a, out);
    return hipGetLastError();
}

hipError_t launch_words_differ_multi(int n, const void* const* a, const void* const* b, const uint64_t* n_bytes, uint32_t* flag,
                                     hipStream_t st) {
    DifferPairs p;
    uint64_t most = 0;
    p.n = n;
    for (int k = 0; k < GDR_DIFFER_MAX; ++k) {
        p.a[k] = k < n ? (const uint32_t*)a[k] : nullptr;
        p.b[k] = k < n ? (const uint32_t*)b[k] : nullptr;
        p.words[k] = k < n ? n_bytes[k] / 4 : 0;
        most = p.words[k] > most ? p.words[k] : most;
    }
    if (most == 0) return hipSuccess;
    const uint64_t n16 = most / 4;
    const int blocks = (int)(n16 / GDR_BLOCK + 1 < 1024 ? n16 / GDR_BLOCK + 1 : 1024);
    hipLaunchKernelGGL(words_differ_multi_kernel, dim3(blocks, n), dim3(GDR_BLOCK), 0, st, p, flag);
    return hipGetLastError();
}

hipError_t launch_words_differ(const void* a, const void* b, uint64_t n_bytes, uint32_t* flag, hipStream_t st) {
    const uint64_t n16 = n_bytes / 16;
    const uint32_t n_tail = (uint32_t)((n_bytes % 16) / 4);
    const int blocks = (int)(n16 / GDR_BLOCK + 1 < 2048 ? n16 / GDR_BLOCK + 1 : 2048);
    hipLaunchKernelGGL(words_differ_kernel, dim3(blocks), dim3(GDR_BLOCK), 0, st, (const uint4*)a, (const uint4*)b, n16,
                       (const uint32_t*)a + 4 * n16, (const uint32_t*)b + 4 * n16, n_tail, flag);
    return hipGetLastError();
}

}  // namespace gdr
