// loss.hip — the image-space loss either side of the render path, fused.
// The adaptor clamps the rendered image to [0,1] (/root/reference/lightning/renderer.py:261) and the
// training loss starts with an MSE against the target (/root/reference/lightning/loss.py:37-38); the
// measurement loss of SURVEY §8d adds 0.1 mean(depth) + 0.1 mean(alpha) so that the depth and alpha
// gradient paths of the rasterizer are exercised.  In torch this is ~12 elementwise / reduction
// launches per view over 5 floats per pixel; here it is one reduction kernel forward and one elementwise
// kernel backward (HBM-bound: 32 B read forward, 12 B read + 20 B written backward, per pixel).
//   loss = mean_{c,p} (clamp(color,0,1) - target)^2 + w_depth mean_p depth + w_alpha mean_p alpha
#include "gdr_common.h"

namespace gdr {
namespace {

__global__ __launch_bounds__(GDR_BLOCK) void view_loss_fwd_kernel(const float* __restrict__ color,
                                                                   const float* __restrict__ depth,
                                                                   const float* __restrict__ alpha,
                                                                   const float* __restrict__ target, int P,
                                                                   float w_depth, float w_alpha,
                                                                   float* __restrict__ loss) {
    __shared__ float wsum[GDR_BLOCK / GDR_WAVE];
    float acc = 0.f;
    const float inv3p = 1.f / (3.f * (float)P), invp = 1.f / (float)P;
    for (int p = blockIdx.x * GDR_BLOCK + threadIdx.x; p < P; p += gridDim.x * GDR_BLOCK) {
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d = fminf(fmaxf(color[(size_t)c * P + p], 0.f), 1.f) - target[(size_t)c * P + p];
            se = fmaf(d, d, se);
        }
        acc += se * inv3p + (w_depth * depth[p] + w_alpha * alpha[p]) * invp;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
}

// d loss / d(color, depth, alpha), scaled by the upstream scalar *g (device memory)
__global__ __launch_bounds__(GDR_BLOCK) void view_loss_bwd_kernel(const float* __restrict__ color,
                                                                   const float* __restrict__ target, int P,
                                                                   float w_depth, float w_alpha,
                                                                   const float* __restrict__ g,
                                                                   float* __restrict__ d_color,
                                                                   float* __restrict__ d_depth,
                                                                   float* __restrict__ d_alpha) {
    const float go = g ? *g : 1.f;
    const float k = go * 2.f / (3.f * (float)P), kd = go * w_depth / (float)P, ka = go * w_alpha / (float)P;
    for (int p = blockIdx.x * GDR_BLOCK + threadIdx.x; p < P; p += gridDim.x * GDR_BLOCK) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = color[(size_t)c * P + p];
            // torch.clamp passes the gradient where min <= x <= max
            const float d = fminf(fmaxf(x, 0.f), 1.f) - target[(size_t)c * P + p];
            d_color[(size_t)c * P + p] = (x >= 0.f && x <= 1.f) ? k * d : 0.f;
        }
        d_depth[p] = kd;
        d_alpha[p] = ka;
    }
}

}  // namespace

hipError_t launch_view_loss_fwd(const float* color, const float* depth, const float* alpha, const float* target,
                                int P, float w_depth, float w_alpha, float* loss, hipStream_t st) {
    const int grid = min(div_up(P, GDR_BLOCK), 2048);
    GDR_LAUNCH(GDR_K_VIEW_LOSS, view_loss_fwd_kernel, dim3(grid), dim3(GDR_BLOCK), st, color, depth, alpha, target, P,
               w_depth, w_alpha, loss);
    return hipGetLastError();
}

hipError_t launch_view_loss_bwd(const float* color, const float* target, int P, float w_depth, float w_alpha,
                                const float* g, float* d_color, float* d_depth, float* d_alpha, hipStream_t st) {
    const int grid = min(div_up(P, GDR_BLOCK), 4096);
    GDR_LAUNCH(GDR_K_VIEW_LOSS, view_loss_bwd_kernel, dim3(grid), dim3(GDR_BLOCK), st, color, target, P, w_depth,
               w_alpha, g, d_color, d_depth, d_alpha);
    return hipGetLastError();
}

}  // namespace gdr
