// select.hip — device top-k of the densification score.
// The reference takes `gradient_point = ||grad[:, 2:4]||_2` of the (N,4) screen-space gradient and turns
// `torch.topk(gradient_point, k_num)` into a boolean mask (/root/reference/lightning/network.py:876-893, k_num =
// 12 000: /root/reference/configs/base.yaml:30); with fewer than k_num points every point is selected.  Only the SET
// matters there, so this is a radix SELECT, not a sort: four 8-bit histogram passes over the float bits of the score
// (scores are >= 0, so the bit pattern orders like the value) find the k-th largest key, one more pass emits the mask
// and, if asked, the indices (unordered).  Ties at the threshold are cut by arrival order (the reference's topk leaves
// their choice unspecified too).  HBM-bound and tiny: 5 passes x 8 B per Gaussian.
// NaN scores follow the reference's two branches (network.py:885-890): with at least k candidates torch.topk ranks NaN
// above every number (key 0x7FFFFFFF here), with fewer than k the mask is `score >= 0`, which is false for NaN.
#include "gdr_common.h"

namespace gdr {
namespace {

struct SelectState {        // lives in the caller's workspace, behind the 256 histogram bins
    uint32_t prefix;        // key bits fixed so far (high bits)
    uint32_t k_rem;         // how many keys with the current prefix are still to be taken
    uint32_t ties_taken;    // emit pass: keys == threshold taken so far
    uint32_t n_out;         // emit pass: indices written so far
    uint32_t fewer;         // pass 0 found fewer than k candidates: the reference's `score >= 0` branch (NaN excluded)
};
#define GDR_KEY_NAN 0x7FFFFFFFu

__device__ __forceinline__ uint32_t score_key(const float* __restrict__ grad, const uint8_t* __restrict__ cand, int i) {
    if (cand && !cand[i]) return 0xFFFFFFFFu;  // not a candidate (never matches a prefix of a real key: see below)
    const float gz = grad[4 * (size_t)i + 2], gw = grad[4 * (size_t)i + 3];
    const float sc = sqrtf(fmaf(gz, gz, gw * gw));
    // >= 0 floats order like their bit patterns; NaN -> the largest key (torch.topk ranks NaN above everything)
    const uint32_t u = __float_as_uint(sc);
    return (sc != sc) ? GDR_KEY_NAN : (u & 0x7FFFFFFFu);
}

// histogram of digit `pass` (0 = most significant byte) of the keys whose higher digits equal the prefix
__global__ __launch_bounds__(GDR_BLOCK) void select_hist_kernel(const float* __restrict__ grad, const uint8_t* __restrict__ cand,
                                                                 int N, int pass, uint32_t* __restrict__ hist,
                                                                 const SelectState* __restrict__ st) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t prefix = pass ? st->prefix : 0u;
    const int shift = 24 - 8 * pass;
    for (int i = blockIdx.x * GDR_BLOCK + threadIdx.x; i < N; i += gridDim.x * GDR_BLOCK) {
        const uint32_t key = score_key(grad, cand, i);
        if (key == 0xFFFFFFFFu) continue;
        if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&h[(key >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// one workgroup: the digit in which the k_rem-th largest key of the current prefix lies; clears the histogram
__global__ __launch_bounds__(GDR_BLOCK) void select_pick_kernel(uint32_t* __restrict__ hist, SelectState* __restrict__ st,
                                                                 int pass, uint32_t k) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = hist[threadIdx.x];
    hist[threadIdx.x] = 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t k_rem = pass ? st->k_rem : k, prefix = pass ? st->prefix : 0u, above = 0u;
        if (pass == 0) {
            uint32_t total = 0u;
            for (int b = 0; b < 256; ++b) total += h[b];
            st->fewer = total < k ? 1u : 0u;
        }
        int d = 255;
        for (; d > 0; --d) {               // largest digit first
            if (above + h[d] >= k_rem) break;
            above += h[d];
        }
        st->prefix = (prefix << 8) | (uint32_t)d;
        st->k_rem = k_rem > above ? k_rem - above : 0u;   // still to take among the keys with this digit
        st->ties_taken = 0u;
        st->n_out = 0u;
    }
}

// keys above the threshold are selected; keys equal to it until k_rem of them are taken
__global__ __launch_bounds__(GDR_BLOCK) void select_emit_kernel(const float* __restrict__ grad, const uint8_t* __restrict__ cand,
                                                                 int N, SelectState* __restrict__ st, uint8_t* __restrict__ mask,
                                                                 int32_t* __restrict__ idx, int k) {
    const uint32_t thr = st->prefix, k_rem = st->k_rem, fewer = st->fewer;
    for (int i = blockIdx.x * GDR_BLOCK + threadIdx.x; i < N; i += gridDim.x * GDR_BLOCK) {
        const uint32_t key = score_key(grad, cand, i);
        bool take = false;
        if (fewer) {   // every candidate with score >= 0 (thr = 0 here: d ran down to 0 in every pass)
            take = key != 0xFFFFFFFFu && key != GDR_KEY_NAN;
        } else if (key != 0xFFFFFFFFu) {
            if (key > thr) take = true;
            else if (key == thr) take = atomicAdd(&st->ties_taken, 1u) < k_rem;
        }
        mask[i] = take ? 1 : 0;
        if (take && idx) {
            const uint32_t o = atomicAdd(&st->n_out, 1u);
            if ((int)o < k) idx[o] = i;
        }
    }
}

}  // namespace

size_t select_workspace_bytes() { return 256 * sizeof(uint32_t) + sizeof(SelectState); }

hipError_t launch_topk_absgrad(int N, const float* grad, const uint8_t* cand, int k, void* workspace,
                               uint8_t* mask, int32_t* idx, hipStream_t st) {
    uint32_t* hist = (uint32_t*)workspace;
    SelectState* state = (SelectState*)(hist + 256);
    hipError_t e = hipMemsetAsync(workspace, 0, select_workspace_bytes(), st);
    if (e != hipSuccess) return e;
    const int blocks = min(div_up(N, GDR_BLOCK), 2048);
    for (int pass = 0; pass < 4; ++pass) {
        GDR_LAUNCH(GDR_K_SELECT, select_hist_kernel, dim3(blocks), dim3(GDR_BLOCK), st, grad, cand, N, pass, hist, state);
        GDR_LAUNCH(GDR_K_SELECT, select_pick_kernel, dim3(1), dim3(GDR_BLOCK), st, hist, state, pass, (uint32_t)k);
    }
    GDR_LAUNCH(GDR_K_SELECT, select_emit_kernel, dim3(blocks), dim3(GDR_BLOCK), st, grad, cand, N, state, mask, idx, k);
    return hipGetLastError();
}

}  // namespace gdr
