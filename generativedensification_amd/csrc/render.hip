// render.hip — K6 (per-tile alpha-composited forward) and K7 (per-pixel reverse-order
// backward) of the rasterizer for gfx950.
// Behaviour: SURVEY.md Appendix A.3 / A.4 (3DGS tile renderer + depth / alpha outputs
// + AbsGS |.|-accumulated screen-space gradients), i.e. what the reference obtains from
// rasterizer(...) at /root/reference/lightning/renderer.py:250-259 and differentiates at
// /root/reference/lightning/network.py:867-878.
//
// CDNA4 mapping (not the 32-wide warp layout of the CUDA lineage):
//   * one workgroup = one 16x16 tile = 4 wavefronts; wave w owns pixel rows 4w..4w+3;
//   * the tile's Gaussian slice is staged 256 entries at a time in LDS as three SoA
//     arrays (xy, conic+opacity, rgb+depth = 40 B/entry) and read back with broadcast
//     ds_read_b64/b128 (all 64 lanes read the same entry: no bank conflicts);
//   * early-out is per WAVE via 64-bit ballots: a wave whose 64 pixels are saturated
//     stops issuing LDS reads; the workgroup stops when all four waves are done;
//   * backward: the 12 per-Gaussian partial gradients are reduced across the 64 lanes
//     with DPP row operations (6 steps) and ONE lane issues the global float atomics
//     (64x fewer atomics than one per pixel); Gaussians no lane of the wave touches are
//     skipped with a single ballot; |.| of the mean2D terms is taken per pixel BEFORE
//     the cross-lane reduction (AbsGS semantics).
//   * blockIdx -> tile mapping is XCD-aware (consecutive tiles share an XCD's L2).
#include "gdr_common.h"

namespace gdr {

namespace {

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// XCD-aware bijective remap of a linear workgroup id (8 XCDs, round-robin dispatch).
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n) {
    const uint32_t xcd = b & 7u, q = n >> 3, r = n & 7u;
    const uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (b >> 3);
}

// alpha evaluation shared by forward and backward: explicit operation order and explicit
// fused multiply-adds so that both kernels make identical skip decisions.
__device__ __forceinline__ float gauss_power(float dx, float dy, float cx, float cy, float cz) {
#pragma clang fp contract(off)
    const float s = fmaf(cz, dy * dy, cx * (dx * dx));
    return fmaf(-0.5f, s, -(cy * (dx * dy)));
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_get(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// sum over the 64 lanes; the total is valid in lane 63.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v += dpp_get<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_get<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += dpp_get<0x141, 0xf>(v);  // row_half_mirror
    v += dpp_get<0x140, 0xf>(v);  // row_mirror   -> every lane: its row-of-16 sum
    v += dpp_get<0x142, 0xa>(v);  // row_bcast15 into rows 1,3
    v += dpp_get<0x143, 0xc>(v);  // row_bcast31 into rows 2,3 -> lane 63 = total
    return v;
}

// ---------------------------------------------------------------------------------
// K6
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(GDR_BLOCK) void render_fwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int gx,
    int ntiles, const float2* __restrict__ xy, const float4* __restrict__ conic_opacity,
    const float4* __restrict__ rgbd, const float* __restrict__ bg, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, float* __restrict__ out_color, float* __restrict__ out_depth,
    float* __restrict__ out_alpha) {
    __shared__ float2 s_xy[GDR_BLOCK];
    __shared__ float4 s_co[GDR_BLOCK];
    __shared__ float4 s_cd[GDR_BLOCK];
    __shared__ int s_done[GDR_BLOCK / GDR_WAVE];

    const uint32_t tile = xcd_remap(blockIdx.x, (uint32_t)ntiles);
    const int tx = (int)(tile % (uint32_t)gx), ty = (int)(tile / (uint32_t)gx);
    const int px = tx * GDR_TILE + (int)(threadIdx.x & 15u);
    const int py = ty * GDR_TILE + (int)(threadIdx.x >> 4);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    int todo = (int)(range.y - range.x);
    const int rounds = (todo + GDR_BLOCK - 1) / GDR_BLOCK;
    const uint32_t wave = threadIdx.x >> 6;

    bool done = !inside;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, Wt = 0.f;
    uint32_t contributor = 0, last_contributor = 0;

    for (int r = 0; r < rounds; ++r, todo -= GDR_BLOCK) {
        // workgroup-level early out: all four waves saturated
        const bool wave_done = __ballot(!done) == 0ull;  // evaluated by all 64 lanes
        if (lane_id() == 0) s_done[wave] = wave_done ? 1 : 0;
        __syncthreads();
        if (s_done[0] + s_done[1] + s_done[2] + s_done[3] == GDR_BLOCK / GDR_WAVE) break;
        const int progress = r * GDR_BLOCK + (int)threadIdx.x;
        if (range.x + progress < range.y) {
            const uint32_t id = point_list[range.x + progress];
            s_xy[threadIdx.x] = xy[id];
            s_co[threadIdx.x] = conic_opacity[id];
            s_cd[threadIdx.x] = rgbd[id];
        }
        __syncthreads();
        const int cnt = todo < GDR_BLOCK ? todo : GDR_BLOCK;
        if (__ballot(!done) != 0ull) {  // wave-uniform: this wave still has live pixels
            for (int j = 0; j < cnt; ++j) {
                if (__ballot(!done) == 0ull) break;  // per-wave early out
                contributor++;
                const float2 m = s_xy[j];
                const float4 co = s_co[j];
                const float dx = m.x - pxf, dy = m.y - pyf;
                const float power = gauss_power(dx, dy, co.x, co.y, co.z);
                if (done || power > 0.f) continue;
                const float alpha = fminf(0.99f, co.w * expf(power));
                if (alpha < (1.f / 255.f)) continue;
                const float test_T = T * (1.f - alpha);
                if (test_T < 0.0001f) {
                    done = true;
                    continue;
                }
                const float4 cd = s_cd[j];
                const float w = alpha * T;
                C0 = fmaf(cd.x, w, C0);
                C1 = fmaf(cd.y, w, C1);
                C2 = fmaf(cd.z, w, C2);
                Dp = fmaf(cd.w, w, Dp);
                Wt += w;
                T = test_T;
                last_contributor = contributor;
            }
        }
    }
    if (inside) {
        const size_t pix = (size_t)py * W + px, P = (size_t)H * W;
        final_T[pix] = T;
        n_contrib[pix] = last_contributor;
        out_color[pix] = fmaf(T, bg[0], C0);
        out_color[P + pix] = fmaf(T, bg[1], C1);
        out_color[2 * P + pix] = fmaf(T, bg[2], C2);
        out_depth[pix] = Dp;
        out_alpha[pix] = Wt;
    }
}

// ---------------------------------------------------------------------------------
// K7
// Accumulators (all pre-zeroed by the launcher):
//   dL_dmean2D (N) float4 : x, y signed (NDC units), z, w = sum |per-pixel term|
//   scratch    (N) 2xfloat4: {dconic.x, dconic.y, dconic.z, ddepth}, {dr, dg, db, -}
//   dL_dopacity(N)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(GDR_BLOCK) void render_bwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int gx,
    int ntiles, const float* __restrict__ bg, const float2* __restrict__ xy,
    const float4* __restrict__ conic_opacity, const float4* __restrict__ rgbd,
    const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dpix, const float* __restrict__ dL_ddepthpix,
    const float* __restrict__ dL_dalphapix, float* __restrict__ dL_dmean2D,
    float* __restrict__ scratch, float* __restrict__ dL_dopacity) {
    __shared__ float2 s_xy[GDR_BLOCK];
    __shared__ float4 s_co[GDR_BLOCK];
    __shared__ float4 s_cd[GDR_BLOCK];
    __shared__ uint32_t s_id[GDR_BLOCK];

    const uint32_t tile = xcd_remap(blockIdx.x, (uint32_t)ntiles);
    const int tx = (int)(tile % (uint32_t)gx), ty = (int)(tile / (uint32_t)gx);
    const int px = tx * GDR_TILE + (int)(threadIdx.x & 15u);
    const int py = ty * GDR_TILE + (int)(threadIdx.x >> 4);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const size_t pix = (size_t)py * W + px, P = (size_t)H * W;
    const uint2 range = ranges[tile];
    int todo = (int)(range.y - range.x);
    const int rounds = (todo + GDR_BLOCK - 1) / GDR_BLOCK;

    const float T_final = inside ? final_T[pix] : 0.f;
    float T = T_final;
    const int last_contributor = inside ? (int)n_contrib[pix] : 0;
    int contributor = todo;
    float gC0 = 0.f, gC1 = 0.f, gC2 = 0.f, gD = 0.f, gA = 0.f;
    if (inside) {
        gC0 = dL_dpix[pix];
        gC1 = dL_dpix[P + pix];
        gC2 = dL_dpix[2 * P + pix];
        if (dL_ddepthpix) gD = dL_ddepthpix[pix];
        if (dL_dalphapix) gA = dL_dalphapix[pix];
    }
    const float bg_dot = (bg[0] * gC0 + bg[1] * gC1) + bg[2] * gC2;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, accD = 0.f, accA = 0.f;
    float last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_depth = 0.f;
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;

    // the deepest contributor any pixel of this wave has: entries behind it are skipped
    int wave_last = last_contributor;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wave_last = max(wave_last, __shfl_xor(wave_last, off, 64));

    for (int r = 0; r < rounds; ++r, todo -= GDR_BLOCK) {
        __syncthreads();
        const int progress = r * GDR_BLOCK + (int)threadIdx.x;
        if (range.x + progress < range.y) {
            const uint32_t id = point_list[range.y - progress - 1];
            s_id[threadIdx.x] = id;
            s_xy[threadIdx.x] = xy[id];
            s_co[threadIdx.x] = conic_opacity[id];
            s_cd[threadIdx.x] = rgbd[id];
        }
        __syncthreads();
        const int cnt = todo < GDR_BLOCK ? todo : GDR_BLOCK;
        if (contributor - cnt >= wave_last) {  // whole chunk lies behind every pixel's last contributor
            contributor -= cnt;
            continue;
        }
        for (int j = 0; j < cnt; ++j) {
            contributor--;
            const float2 m = s_xy[j];
            const float4 co = s_co[j];
            const float dx = m.x - pxf, dy = m.y - pyf;
            const float power = gauss_power(dx, dy, co.x, co.y, co.z);
            const float G = expf(power);
            const float alpha = fminf(0.99f, co.w * G);
            const bool hit = (contributor < last_contributor) && !(power > 0.f) && !(alpha < (1.f / 255.f));
            if (__ballot(hit) == 0ull) continue;  // wave-uniform skip

            float v_mx = 0.f, v_my = 0.f, v_ax = 0.f, v_ay = 0.f, v_cx = 0.f, v_cy = 0.f, v_cz = 0.f;
            float v_dd = 0.f, v_r = 0.f, v_g = 0.f, v_b = 0.f, v_o = 0.f;
            if (hit) {
                const float4 cd = s_cd[j];
                T = T / (1.f - alpha);
                const float w = alpha * T;
                acc0 = last_alpha * lc0 + (1.f - last_alpha) * acc0;
                acc1 = last_alpha * lc1 + (1.f - last_alpha) * acc1;
                acc2 = last_alpha * lc2 + (1.f - last_alpha) * acc2;
                lc0 = cd.x; lc1 = cd.y; lc2 = cd.z;
                float dL_dalpha = (cd.x - acc0) * gC0 + (cd.y - acc1) * gC1 + (cd.z - acc2) * gC2;
                v_r = w * gC0; v_g = w * gC1; v_b = w * gC2;
                accD = last_alpha * last_depth + (1.f - last_alpha) * accD;
                last_depth = cd.w;
                dL_dalpha += (cd.w - accD) * gD;
                v_dd = w * gD;
                accA = last_alpha + (1.f - last_alpha) * accA;
                dL_dalpha += (1.f - accA) * gA;
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                const float dL_dG = co.w * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * co.x - gdy * co.y;
                const float dG_ddely = -gdy * co.z - gdx * co.y;
                v_mx = dL_dG * dG_ddelx * ddelx_dx;
                v_my = dL_dG * dG_ddely * ddely_dy;
                v_ax = fabsf(v_mx);
                v_ay = fabsf(v_my);
                v_cx = -0.5f * gdx * dx * dL_dG;
                v_cy = -gdx * dy * dL_dG;
                v_cz = -0.5f * gdy * dy * dL_dG;
                v_o = G * dL_dalpha;
            }
            v_mx = wave_sum_to_lane63(v_mx); v_my = wave_sum_to_lane63(v_my);
            v_ax = wave_sum_to_lane63(v_ax); v_ay = wave_sum_to_lane63(v_ay);
            v_cx = wave_sum_to_lane63(v_cx); v_cy = wave_sum_to_lane63(v_cy);
            v_cz = wave_sum_to_lane63(v_cz); v_dd = wave_sum_to_lane63(v_dd);
            v_r = wave_sum_to_lane63(v_r);   v_g = wave_sum_to_lane63(v_g);
            v_b = wave_sum_to_lane63(v_b);   v_o = wave_sum_to_lane63(v_o);
            if (lane_id() == 63) {
                const uint32_t id = s_id[j];
                float* m2 = dL_dmean2D + 4 * (size_t)id;
                float* sc = scratch + 8 * (size_t)id;
                atomicAdd(m2 + 0, v_mx); atomicAdd(m2 + 1, v_my);
                atomicAdd(m2 + 2, v_ax); atomicAdd(m2 + 3, v_ay);
                atomicAdd(sc + 0, v_cx); atomicAdd(sc + 1, v_cy);
                atomicAdd(sc + 2, v_cz); atomicAdd(sc + 3, v_dd);
                atomicAdd(sc + 4, v_r);  atomicAdd(sc + 5, v_g);
                atomicAdd(sc + 6, v_b);
                atomicAdd(dL_dopacity + id, v_o);
            }
        }
    }
}

}  // namespace

hipError_t launch_render_fwd(const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                             const gdr_image* img, const gdr_outputs* out, hipStream_t st) {
    const int W = s->image_width, H = s->image_height;
    const int gx = tile_grid_x(W), gy = tile_grid_y(H);
    const int ntiles = gx * gy;
    GDR_LAUNCH(GDR_K_RENDER_FWD, render_fwd_kernel, dim3(ntiles), dim3(GDR_BLOCK), st,
                       (const uint2*)img->ranges, bin->values[bin->sorted], W, H, gx, ntiles,
                       (const float2*)g->xy, (const float4*)g->conic_opacity, (const float4*)g->rgb,
                       s->bg, img->final_T, img->n_contrib, out->color, out->depth, out->alpha);
    return hipGetLastError();
}

hipError_t launch_render_bwd(const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                             const gdr_image* img, const gdr_grad_inputs* gi,
                             const gdr_grad_outputs* go, hipStream_t st) {
    const int W = s->image_width, H = s->image_height;
    const int gx = tile_grid_x(W), gy = tile_grid_y(H);
    const int ntiles = gx * gy;
    GDR_LAUNCH(GDR_K_RENDER_BWD, render_bwd_kernel, dim3(ntiles), dim3(GDR_BLOCK), st,
                       (const uint2*)img->ranges, bin->values[bin->sorted], W, H, gx, ntiles, s->bg,
                       (const float2*)g->xy, (const float4*)g->conic_opacity, (const float4*)g->rgb,
                       img->final_T, img->n_contrib, gi->dL_dcolor, gi->dL_ddepth, gi->dL_dalpha,
                       go->dL_dmeans2D, go->scratch, go->dL_dopacities);
    return hipGetLastError();
}

}  // namespace gdr
