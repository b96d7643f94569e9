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
#include <stdlib.h>

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


// =================================================================================
// v2 kernels: sub-tile culling.  Each wave owns an 8x8 pixel sub-tile of the 16x16 tile.
// While a 256-entry slice of the tile's sorted list sits in LDS, every lane tests ONE
// entry's alpha >= 1/255 bounding box against the wave's sub-tile (4 ballots cover the
// slice); only the surviving entries are evaluated, in list order, via a scalar
// find-first-set loop over the 64-bit masks.  The test is conservative (exact bbox of the
// alpha >= 1/255 ellipse, widened), so results — including n_contrib, which stays the
// 1-based position in the FULL tile list — are identical to evaluating every entry.
// The conic is pre-multiplied by log2(e) when staged so that G = v_exp_f32(power) with no
// range reduction; forward and backward stage identically => identical skip decisions.
// The next slice is fetched into registers while the current one is being composited.
// =================================================================================
#define GDR_LOG2E 1.4426950408889634f
#define GDR_LN2 0.6931471805599453f

struct Staged {
    float2 xy, ext;
    float4 co, cd;
};

// conservative half-extent (pixels) of {alpha >= 1/255} for conic (cx,cy,cz) and opacity o
__device__ __forceinline__ float2 alpha_extent(const float4 co) {
    const float t = 255.f * co.w;
    if (!(t > 1.f)) return make_float2(-1.f, -1.f);  // alpha <= o < 1/255 everywhere (also NaN)
    const float det = co.x * co.z - co.y * co.y;
    if (!(det > 0.f)) return make_float2(1e30f, 1e30f);  // degenerate: never cull
    const float tau2 = 2.f * __logf(t) / det;            // 2 ln(255 o) / det(conic)
    return make_float2(sqrtf(tau2 * co.z) * 1.002f + 0.02f, sqrtf(tau2 * co.x) * 1.002f + 0.02f);
}

__device__ __forceinline__ Staged stage_entry(float2 xy, float4 co, float4 cd) {
    Staged s;
    s.xy = xy;
    s.ext = alpha_extent(co);
    s.co = make_float4(co.x * GDR_LOG2E, co.y * GDR_LOG2E, co.z * GDR_LOG2E, co.w);
    s.cd = cd;
    return s;
}

__device__ __forceinline__ uint64_t cull_mask(const float2* s_xy, const float2* s_ext, int e, float X0,
                                              float X1, float Y0, float Y1) {
    const float2 m = s_xy[e];
    const float2 h = s_ext[e];
    const bool ov = (h.x >= 0.f) && (m.x + h.x >= X0) && (m.x - h.x <= X1) && (m.y + h.y >= Y0) &&
                    (m.y - h.y <= Y1);
    return __ballot(ov);
}

__global__ __launch_bounds__(GDR_BLOCK) void render_fwd_v2_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int gx,
    int ntiles, const float2* __restrict__ xy, const float4* __restrict__ conic_opacity,
    const float4* __restrict__ rgbd, const float* __restrict__ bg, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, float* __restrict__ out_color, float* __restrict__ out_depth,
    float* __restrict__ out_alpha) {
    __shared__ float2 s_xy[GDR_BLOCK];
    __shared__ float2 s_ext[GDR_BLOCK];
    __shared__ float4 s_co[GDR_BLOCK];
    __shared__ float4 s_cd[GDR_BLOCK];
    __shared__ int s_done[GDR_BLOCK / GDR_WAVE];

    const uint32_t tile = xcd_remap(blockIdx.x, (uint32_t)ntiles);
    const int tx = (int)(tile % (uint32_t)gx), ty = (int)(tile / (uint32_t)gx);
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const int sx0 = tx * GDR_TILE + (int)(wave & 1u) * 8, sy0 = ty * GDR_TILE + (int)(wave >> 1) * 8;
    const int px = sx0 + (int)(lane & 7u), py = sy0 + (int)(lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float X0 = (float)sx0, X1 = (float)(sx0 + 7), Y0 = (float)sy0, Y1 = (float)(sy0 + 7);
    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    const int rounds = (total + GDR_BLOCK - 1) / GDR_BLOCK;

    bool done = !inside;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, Wt = 0.f;
    uint32_t last_contributor = 0;

    float2 r_xy = make_float2(0.f, 0.f);
    float4 r_co = make_float4(0.f, 0.f, 0.f, 0.f), r_cd = r_co;
    bool r_valid = (int)threadIdx.x < total;
    if (r_valid) {
        const uint32_t id = point_list[range.x + threadIdx.x];
        r_xy = xy[id]; r_co = conic_opacity[id]; r_cd = rgbd[id];
    }
    for (int r = 0; r < rounds; ++r) {
        const bool wave_done = __ballot(!done) == 0ull;
        if (lane == 0) s_done[wave] = wave_done ? 1 : 0;
        __syncthreads();
        if (s_done[0] + s_done[1] + s_done[2] + s_done[3] == GDR_BLOCK / GDR_WAVE) break;
        if (r_valid) {
            const Staged st = stage_entry(r_xy, r_co, r_cd);
            s_xy[threadIdx.x] = st.xy; s_ext[threadIdx.x] = st.ext;
            s_co[threadIdx.x] = st.co; s_cd[threadIdx.x] = st.cd;
        } else {
            s_ext[threadIdx.x] = make_float2(-1.f, -1.f);
            s_xy[threadIdx.x] = make_float2(0.f, 0.f);
        }
        __syncthreads();
        {   // prefetch the next slice (lands while this one is composited)
            const int nxt = (r + 1) * GDR_BLOCK + (int)threadIdx.x;
            r_valid = nxt < total;
            if (r_valid) {
                const uint32_t id = point_list[range.x + nxt];
                r_xy = xy[id]; r_co = conic_opacity[id]; r_cd = rgbd[id];
            }
        }
        if (wave_done) continue;
        const uint32_t base = (uint32_t)(r * GDR_BLOCK);
#pragma unroll 1
        for (int g = 0; g < GDR_BLOCK / GDR_WAVE; ++g) {
            uint64_t mask = cull_mask(s_xy, s_ext, g * GDR_WAVE + (int)lane, X0, X1, Y0, Y1);
            while (mask) {
                const int e = g * GDR_WAVE + __builtin_ctzll(mask);
                mask &= mask - 1ull;
                const float2 m = s_xy[e];
                const float4 co = s_co[e];
                const float dx = m.x - pxf, dy = m.y - pyf;
                const float p2 = gauss_power(dx, dy, co.x, co.y, co.z);
                const float alpha = fminf(0.99f, co.w * __builtin_amdgcn_exp2f(p2));
                const bool c = !done && !(p2 > 0.f) && !(alpha < (1.f / 255.f));
                if (__ballot(c) == 0ull) continue;
                const float test_T = T * (1.f - alpha);
                const bool stop = c && (test_T < 0.0001f);
                done = done || stop;
                const bool acc = c && !stop;
                const float4 cd = s_cd[e];
                const float w = acc ? alpha * T : 0.f;
                C0 = fmaf(cd.x, w, C0);
                C1 = fmaf(cd.y, w, C1);
                C2 = fmaf(cd.z, w, C2);
                Dp = fmaf(cd.w, w, Dp);
                Wt += w;
                T = acc ? test_T : T;
                last_contributor = acc ? base + (uint32_t)e + 1u : last_contributor;
                if (__ballot(!done) == 0ull) { mask = 0ull; g = GDR_BLOCK / GDR_WAVE; }
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

#ifndef GDR_ABL
#define GDR_ABL 0
#endif
#define GDR_ACC_STRIDE 13  // 12 partial gradients + touched flag; odd stride: conflict-free per-thread rows

__global__ __launch_bounds__(GDR_BLOCK) void render_bwd_v2_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int gx,
    int ntiles, const float* __restrict__ bg, const float2* __restrict__ xy,
    const float4* __restrict__ conic_opacity, const float4* __restrict__ rgbd,
    const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dpix, const float* __restrict__ dL_ddepthpix,
    const float* __restrict__ dL_dalphapix, float* __restrict__ dL_dmean2D,
    float* __restrict__ scratch, float* __restrict__ dL_dopacity) {
    __shared__ float2 s_xy[GDR_BLOCK];
    __shared__ float2 s_ext[GDR_BLOCK];
    __shared__ float4 s_co[GDR_BLOCK];
    __shared__ float4 s_cd[GDR_BLOCK];
    __shared__ uint32_t s_id[GDR_BLOCK];
    __shared__ float s_acc[GDR_BLOCK * GDR_ACC_STRIDE];

    const uint32_t tile = xcd_remap(blockIdx.x, (uint32_t)ntiles);
    const int tx = (int)(tile % (uint32_t)gx), ty = (int)(tile / (uint32_t)gx);
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const int sx0 = tx * GDR_TILE + (int)(wave & 1u) * 8, sy0 = ty * GDR_TILE + (int)(wave >> 1) * 8;
    const int px = sx0 + (int)(lane & 7u), py = sy0 + (int)(lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float X0 = (float)sx0, X1 = (float)(sx0 + 7), Y0 = (float)sy0, Y1 = (float)(sy0 + 7);
    const size_t pix = (size_t)py * W + px, P = (size_t)H * W;
    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    const int rounds = (total + GDR_BLOCK - 1) / GDR_BLOCK;

    const float T_final = inside ? final_T[pix] : 0.f;
    float T = T_final;
    const int last_contributor = inside ? (int)n_contrib[pix] : 0;
    float gC0 = 0.f, gC1 = 0.f, gC2 = 0.f, gD = 0.f, gA = 0.f;
    if (inside) {
        gC0 = dL_dpix[pix]; gC1 = dL_dpix[P + pix]; gC2 = dL_dpix[2 * P + pix];
        if (dL_ddepthpix) gD = dL_ddepthpix[pix];
        if (dL_dalphapix) gA = dL_dalphapix[pix];
    }
    const float bg_dot = (bg[0] * gC0 + bg[1] * gC1) + bg[2] * gC2;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, accD = 0.f, accA = 0.f;
    float last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_depth = 0.f;
    // the staged conic is log2(e) x the true one: fold 1/log2(e) = ln 2 into the pixel->NDC factors
    const float kx = 0.5f * (float)W * GDR_LN2, ky = 0.5f * (float)H * GDR_LN2;

    int wave_last = last_contributor;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wave_last = max(wave_last, __shfl_xor(wave_last, off, 64));

    for (int k = 0; k < GDR_ACC_STRIDE; ++k) s_acc[threadIdx.x * GDR_ACC_STRIDE + k] = 0.f;

    // slice r holds list positions total-1-(r*256+e), e = 0..255: back to front
    float2 r_xy = make_float2(0.f, 0.f);
    float4 r_co = make_float4(0.f, 0.f, 0.f, 0.f), r_cd = r_co;
    uint32_t r_id = 0;
    bool r_valid = (int)threadIdx.x < total;
    if (r_valid) {
        r_id = point_list[range.y - 1 - threadIdx.x];
        r_xy = xy[r_id]; r_co = conic_opacity[r_id]; r_cd = rgbd[r_id];
    }
    for (int r = 0; r <= rounds; ++r) {
        __syncthreads();  // every wave finished the previous slice: its accumulators are complete
        if (r > 0) {      // flush my entry of the previous slice: one coalesced atomic pass
            float* a = s_acc + threadIdx.x * GDR_ACC_STRIDE;
            if (a[12] != 0.f) {
                const uint32_t id = s_id[threadIdx.x];
                float* m2 = dL_dmean2D + 4 * (size_t)id;
                float* sc = scratch + 8 * (size_t)id;
                atomicAdd(m2 + 0, a[0]); atomicAdd(m2 + 1, a[1]); atomicAdd(m2 + 2, a[2]); atomicAdd(m2 + 3, a[3]);
                atomicAdd(sc + 0, a[4]); atomicAdd(sc + 1, a[5]); atomicAdd(sc + 2, a[6]); atomicAdd(sc + 3, a[7]);
                atomicAdd(sc + 4, a[8]); atomicAdd(sc + 5, a[9]); atomicAdd(sc + 6, a[10]);
                atomicAdd(dL_dopacity + id, a[11]);
#pragma unroll
                for (int k = 0; k < GDR_ACC_STRIDE; ++k) a[k] = 0.f;
            }
        }
        if (r == rounds) break;
        if (r_valid) {
            const Staged st = stage_entry(r_xy, r_co, r_cd);
            s_xy[threadIdx.x] = st.xy; s_ext[threadIdx.x] = st.ext;
            s_co[threadIdx.x] = st.co; s_cd[threadIdx.x] = st.cd;
            s_id[threadIdx.x] = r_id;
        } else {
            s_ext[threadIdx.x] = make_float2(-1.f, -1.f);
            s_xy[threadIdx.x] = make_float2(0.f, 0.f);
        }
        __syncthreads();
        {
            const int nxt = (r + 1) * GDR_BLOCK + (int)threadIdx.x;
            r_valid = nxt < total;
            if (r_valid) {
                r_id = point_list[range.y - 1 - nxt];
                r_xy = xy[r_id]; r_co = conic_opacity[r_id]; r_cd = rgbd[r_id];
            }
        }
        // list position of LDS entry e in this slice: pos = top - e
        const int top = total - 1 - r * GDR_BLOCK;
        if (top - (GDR_BLOCK - 1) >= wave_last) continue;  // whole slice behind every pixel's last contributor
#pragma unroll 1
        for (int g = 0; g < GDR_BLOCK / GDR_WAVE; ++g) {
            uint64_t mask = cull_mask(s_xy, s_ext, g * GDR_WAVE + (int)lane, X0, X1, Y0, Y1);
            while (mask) {
                const int e = g * GDR_WAVE + __builtin_ctzll(mask);
                mask &= mask - 1ull;
                const int pos = top - e;
                const float2 m = s_xy[e];
                const float4 co = s_co[e];
                const float dx = m.x - pxf, dy = m.y - pyf;
                const float p2 = gauss_power(dx, dy, co.x, co.y, co.z);
                const float G = __builtin_amdgcn_exp2f(p2);
                const float alpha = fminf(0.99f, co.w * G);
                const bool hit = (pos < last_contributor) && !(p2 > 0.f) && !(alpha < (1.f / 255.f));
                if (__ballot(hit) == 0ull) continue;
                const float4 cd = s_cd[e];
                const float oma = 1.f - alpha;
                const float Tn = T / oma;
                const float w = hit ? alpha * Tn : 0.f;
                const float n0 = last_alpha * lc0 + (1.f - last_alpha) * acc0;
                const float n1 = last_alpha * lc1 + (1.f - last_alpha) * acc1;
                const float n2 = last_alpha * lc2 + (1.f - last_alpha) * acc2;
                const float nD = last_alpha * last_depth + (1.f - last_alpha) * accD;
                const float nA = last_alpha + (1.f - last_alpha) * accA;
                float dL_dalpha = (cd.x - n0) * gC0 + (cd.y - n1) * gC1 + (cd.z - n2) * gC2;
                dL_dalpha += (cd.w - nD) * gD;
                dL_dalpha += (1.f - nA) * gA;
                dL_dalpha *= Tn;
                dL_dalpha += (-T_final / oma) * bg_dot;
                dL_dalpha = hit ? dL_dalpha : 0.f;
                if (hit) {
                    T = Tn; acc0 = n0; acc1 = n1; acc2 = n2; accD = nD; accA = nA;
                    lc0 = cd.x; lc1 = cd.y; lc2 = cd.z; last_depth = cd.w; last_alpha = alpha;
                }
                const float dL_dG = co.w * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                // co.xyz carry the log2(e) factor; kx, ky carry its inverse
                float v_mx = dL_dG * (-gdx * co.x - gdy * co.y) * kx;
                float v_my = dL_dG * (-gdy * co.z - gdx * co.y) * ky;
                float v_ax = fabsf(v_mx), v_ay = fabsf(v_my);
                float v_cx = -0.5f * gdx * dx * dL_dG;
                float v_cy = -gdx * dy * dL_dG;
                float v_cz = -0.5f * gdy * dy * dL_dG;
                float v_dd = w * gD, v_r = w * gC0, v_g = w * gC1, v_b = w * gC2;
                float v_o = G * dL_dalpha;
#if GDR_ABL != 2
                v_mx = wave_sum_to_lane63(v_mx); v_my = wave_sum_to_lane63(v_my);
                v_ax = wave_sum_to_lane63(v_ax); v_ay = wave_sum_to_lane63(v_ay);
                v_cx = wave_sum_to_lane63(v_cx); v_cy = wave_sum_to_lane63(v_cy);
                v_cz = wave_sum_to_lane63(v_cz); v_dd = wave_sum_to_lane63(v_dd);
                v_r = wave_sum_to_lane63(v_r);   v_g = wave_sum_to_lane63(v_g);
                v_b = wave_sum_to_lane63(v_b);   v_o = wave_sum_to_lane63(v_o);
#endif
#if GDR_ABL == 1 || GDR_ABL == 3
                asm volatile("" ::"v"(v_mx), "v"(v_my), "v"(v_ax), "v"(v_ay), "v"(v_cx), "v"(v_cy));
                asm volatile("" ::"v"(v_cz), "v"(v_dd), "v"(v_r), "v"(v_g), "v"(v_b), "v"(v_o));
                if (false) {
#else
                if (lane == 63) {
#endif
                    float* a = s_acc + e * GDR_ACC_STRIDE;
                    atomicAdd(a + 0, v_mx); atomicAdd(a + 1, v_my); atomicAdd(a + 2, v_ax); atomicAdd(a + 3, v_ay);
                    atomicAdd(a + 4, v_cx); atomicAdd(a + 5, v_cy); atomicAdd(a + 6, v_cz); atomicAdd(a + 7, v_dd);
                    atomicAdd(a + 8, v_r);  atomicAdd(a + 9, v_g);  atomicAdd(a + 10, v_b); atomicAdd(a + 11, v_o);
                    a[12] = 1.f;
                }
            }
        }
    }
}


// =================================================================================
// v3 kernels: 4x4-pixel blocks, one per 16-lane DPP row.
// A wave still owns an 8x8 sub-tile, but each of its four rows of 16 lanes composites
// its OWN 4x4 block against its OWN culled sub-list, so one wave instruction advances up
// to four different Gaussians.  For footprints of a few pixels (the densified regime)
// this cuts the evaluated pixel-Gaussian pairs ~2.3x versus 8x8 culling, and the
// backward's cross-lane reduction shrinks from 6 DPP steps over 64 lanes to 4 row-local
// steps shared by four Gaussians.  Per-block masks live in SGPRs (one ballot per block
// per 64 staged entries); the per-lane entry index is a select over four scalars.
// =================================================================================
struct RowPick {
    int e[4];
};

__device__ __forceinline__ int pick_next(uint64_t& m) {
    if (m == 0ull) return 64;
    const int b = __builtin_ctzll(m);
    m &= m - 1ull;
    return b;
}

// sum over each row of 16 lanes; every lane of the row receives its row's total
__device__ __forceinline__ float row_sum(float v) {
    v += dpp_get<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_get<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += dpp_get<0x141, 0xf>(v);  // row_half_mirror
    v += dpp_get<0x140, 0xf>(v);  // row_mirror
    return v;
}

#define GDR_ROW_MASK(k) (0xFFFFull << (16 * (k)))

__global__ __launch_bounds__(GDR_BLOCK) void render_fwd_v3_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int gx,
    int ntiles, const float2* __restrict__ xy, const float4* __restrict__ conic_opacity,
    const float4* __restrict__ rgbd, const float* __restrict__ bg, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, float* __restrict__ out_color, float* __restrict__ out_depth,
    float* __restrict__ out_alpha) {
    __shared__ float2 s_xy[GDR_BLOCK];
    __shared__ float2 s_ext[GDR_BLOCK];
    __shared__ float4 s_co[GDR_BLOCK];
    __shared__ float4 s_cd[GDR_BLOCK];
    __shared__ int s_done[GDR_BLOCK / GDR_WAVE];

    const uint32_t tile = xcd_remap(blockIdx.x, (uint32_t)ntiles);
    const int tx = (int)(tile % (uint32_t)gx), ty = (int)(tile / (uint32_t)gx);
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t row = lane >> 4, li = lane & 15u;
    const int sx0 = tx * GDR_TILE + (int)(wave & 1u) * 8, sy0 = ty * GDR_TILE + (int)(wave >> 1) * 8;
    const int px = sx0 + (int)(row & 1u) * 4 + (int)(li & 3u), py = sy0 + (int)(row >> 1) * 4 + (int)(li >> 2);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float XA = (float)sx0, YA = (float)sy0;  // block k: x in [XA+4(k&1), +3], y in [YA+4(k>>1), +3]
    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    const int rounds = (total + GDR_BLOCK - 1) / GDR_BLOCK;

    bool done = !inside;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, Wt = 0.f;
    uint32_t last_contributor = 0;

    float2 r_xy = make_float2(0.f, 0.f);
    float4 r_co = make_float4(0.f, 0.f, 0.f, 0.f), r_cd = r_co;
    bool r_valid = (int)threadIdx.x < total;
    if (r_valid) {
        const uint32_t id = point_list[range.x + threadIdx.x];
        r_xy = xy[id]; r_co = conic_opacity[id]; r_cd = rgbd[id];
    }
    for (int r = 0; r < rounds; ++r) {
        uint64_t live = __ballot(!done);
        if (lane == 0) s_done[wave] = live == 0ull ? 1 : 0;
        __syncthreads();
        if (s_done[0] + s_done[1] + s_done[2] + s_done[3] == GDR_BLOCK / GDR_WAVE) break;
        if (r_valid) {
            const Staged st = stage_entry(r_xy, r_co, r_cd);
            s_xy[threadIdx.x] = st.xy; s_ext[threadIdx.x] = st.ext;
            s_co[threadIdx.x] = st.co; s_cd[threadIdx.x] = st.cd;
        } else {
            s_ext[threadIdx.x] = make_float2(-1.f, -1.f);
            s_xy[threadIdx.x] = make_float2(0.f, 0.f);
        }
        __syncthreads();
        {
            const int nxt = (r + 1) * GDR_BLOCK + (int)threadIdx.x;
            r_valid = nxt < total;
            if (r_valid) {
                const uint32_t id = point_list[range.x + nxt];
                r_xy = xy[id]; r_co = conic_opacity[id]; r_cd = rgbd[id];
            }
        }
        if (live == 0ull) continue;
        const uint32_t base = (uint32_t)(r * GDR_BLOCK);
#pragma unroll 1
        for (int g = 0; g < GDR_BLOCK / GDR_WAVE; ++g) {
            uint64_t m0, m1, m2, m3;
            {
                const float2 m = s_xy[g * GDR_WAVE + (int)lane];
                const float2 h = s_ext[g * GDR_WAVE + (int)lane];
                const bool v = h.x >= 0.f;
                const float lo_x = m.x - h.x, hi_x = m.x + h.x, lo_y = m.y - h.y, hi_y = m.y + h.y;
                const bool x0 = v && hi_x >= XA && lo_x <= XA + 3.f, x1 = v && hi_x >= XA + 4.f && lo_x <= XA + 7.f;
                const bool y0 = hi_y >= YA && lo_y <= YA + 3.f, y1 = hi_y >= YA + 4.f && lo_y <= YA + 7.f;
                m0 = (live & GDR_ROW_MASK(0)) ? __ballot(x0 && y0) : 0ull;
                m1 = (live & GDR_ROW_MASK(1)) ? __ballot(x1 && y0) : 0ull;
                m2 = (live & GDR_ROW_MASK(2)) ? __ballot(x0 && y1) : 0ull;
                m3 = (live & GDR_ROW_MASK(3)) ? __ballot(x1 && y1) : 0ull;
            }
            while ((m0 | m1 | m2 | m3) != 0ull) {
                const int e0 = pick_next(m0), e1 = pick_next(m1), e2 = pick_next(m2), e3 = pick_next(m3);
                const int es = row == 0 ? e0 : (row == 1 ? e1 : (row == 2 ? e2 : e3));
                const bool act = es < 64;
                const int e = g * GDR_WAVE + (act ? es : 0);
                const float2 m = s_xy[e];
                const float4 co = s_co[e];
                const float dx = m.x - pxf, dy = m.y - pyf;
                const float p2 = gauss_power(dx, dy, co.x, co.y, co.z);
                const float alpha = fminf(0.99f, co.w * __builtin_amdgcn_exp2f(p2));
                const bool c = act && !done && !(p2 > 0.f) && !(alpha < (1.f / 255.f));
                if (__ballot(c) == 0ull) continue;
                const float test_T = T * (1.f - alpha);
                const bool stop = c && (test_T < 0.0001f);
                done = done || stop;
                const bool acc = c && !stop;
                const float4 cd = s_cd[e];
                const float w = acc ? alpha * T : 0.f;
                C0 = fmaf(cd.x, w, C0);
                C1 = fmaf(cd.y, w, C1);
                C2 = fmaf(cd.z, w, C2);
                Dp = fmaf(cd.w, w, Dp);
                Wt += w;
                T = acc ? test_T : T;
                last_contributor = acc ? base + (uint32_t)e + 1u : last_contributor;
                if (__ballot(stop) != 0ull) {  // rare: some pixel saturated -> retire finished blocks
                    live = __ballot(!done);
                    if (!(live & GDR_ROW_MASK(0))) m0 = 0ull;
                    if (!(live & GDR_ROW_MASK(1))) m1 = 0ull;
                    if (!(live & GDR_ROW_MASK(2))) m2 = 0ull;
                    if (!(live & GDR_ROW_MASK(3))) m3 = 0ull;
                    if (live == 0ull) g = GDR_BLOCK / GDR_WAVE;
                }
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

__global__ __launch_bounds__(GDR_BLOCK) void render_bwd_v3_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int gx,
    int ntiles, const float* __restrict__ bg, const float2* __restrict__ xy,
    const float4* __restrict__ conic_opacity, const float4* __restrict__ rgbd,
    const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dpix, const float* __restrict__ dL_ddepthpix,
    const float* __restrict__ dL_dalphapix, float* __restrict__ dL_dmean2D,
    float* __restrict__ scratch, float* __restrict__ dL_dopacity) {
    __shared__ float2 s_xy[GDR_BLOCK];
    __shared__ float2 s_ext[GDR_BLOCK];
    __shared__ float4 s_co[GDR_BLOCK];
    __shared__ float4 s_cd[GDR_BLOCK];
    __shared__ uint32_t s_id[GDR_BLOCK];
    __shared__ float s_acc[GDR_BLOCK * GDR_ACC_STRIDE];

    const uint32_t tile = xcd_remap(blockIdx.x, (uint32_t)ntiles);
    const int tx = (int)(tile % (uint32_t)gx), ty = (int)(tile / (uint32_t)gx);
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t row = lane >> 4, li = lane & 15u;
    const int sx0 = tx * GDR_TILE + (int)(wave & 1u) * 8, sy0 = ty * GDR_TILE + (int)(wave >> 1) * 8;
    const int px = sx0 + (int)(row & 1u) * 4 + (int)(li & 3u), py = sy0 + (int)(row >> 1) * 4 + (int)(li >> 2);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float XA = (float)sx0, YA = (float)sy0;
    const size_t pix = (size_t)py * W + px, P = (size_t)H * W;
    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    const int rounds = (total + GDR_BLOCK - 1) / GDR_BLOCK;

    const float T_final = inside ? final_T[pix] : 0.f;
    float T = T_final;
    const int last_contributor = inside ? (int)n_contrib[pix] : 0;
    float gC0 = 0.f, gC1 = 0.f, gC2 = 0.f, gD = 0.f, gA = 0.f;
    if (inside) {
        gC0 = dL_dpix[pix]; gC1 = dL_dpix[P + pix]; gC2 = dL_dpix[2 * P + pix];
        if (dL_ddepthpix) gD = dL_ddepthpix[pix];
        if (dL_dalphapix) gA = dL_dalphapix[pix];
    }
    const float bg_dot = (bg[0] * gC0 + bg[1] * gC1) + bg[2] * gC2;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, accD = 0.f, accA = 0.f;
    float last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_depth = 0.f;
    const float kx = 0.5f * (float)W * GDR_LN2, ky = 0.5f * (float)H * GDR_LN2;

    // deepest contributor per 4x4 block (row of 16 lanes) and per wave
    int row_last = last_contributor;
    row_last = max(row_last, __shfl_xor(row_last, 1, 64));
    row_last = max(row_last, __shfl_xor(row_last, 2, 64));
    row_last = max(row_last, __shfl_xor(row_last, 4, 64));
    row_last = max(row_last, __shfl_xor(row_last, 8, 64));
    const int rl0 = __builtin_amdgcn_readlane(row_last, 0), rl1 = __builtin_amdgcn_readlane(row_last, 16);
    const int rl2 = __builtin_amdgcn_readlane(row_last, 32), rl3 = __builtin_amdgcn_readlane(row_last, 48);
    const int wave_last = max(max(rl0, rl1), max(rl2, rl3));

    for (int k = 0; k < GDR_ACC_STRIDE; ++k) s_acc[threadIdx.x * GDR_ACC_STRIDE + k] = 0.f;

    float2 r_xy = make_float2(0.f, 0.f);
    float4 r_co = make_float4(0.f, 0.f, 0.f, 0.f), r_cd = r_co;
    uint32_t r_id = 0;
    bool r_valid = (int)threadIdx.x < total;
    if (r_valid) {
        r_id = point_list[range.y - 1 - threadIdx.x];
        r_xy = xy[r_id]; r_co = conic_opacity[r_id]; r_cd = rgbd[r_id];
    }
    for (int r = 0; r <= rounds; ++r) {
        __syncthreads();
        if (r > 0) {
            float* a = s_acc + threadIdx.x * GDR_ACC_STRIDE;
            if (a[12] != 0.f) {
                const uint32_t id = s_id[threadIdx.x];
                float* m2 = dL_dmean2D + 4 * (size_t)id;
                float* sc = scratch + 8 * (size_t)id;
                atomicAdd(m2 + 0, a[0]); atomicAdd(m2 + 1, a[1]); atomicAdd(m2 + 2, a[2]); atomicAdd(m2 + 3, a[3]);
                atomicAdd(sc + 0, a[4]); atomicAdd(sc + 1, a[5]); atomicAdd(sc + 2, a[6]); atomicAdd(sc + 3, a[7]);
                atomicAdd(sc + 4, a[8]); atomicAdd(sc + 5, a[9]); atomicAdd(sc + 6, a[10]);
                atomicAdd(dL_dopacity + id, a[11]);
#pragma unroll
                for (int k = 0; k < GDR_ACC_STRIDE; ++k) a[k] = 0.f;
            }
        }
        if (r == rounds) break;
        if (r_valid) {
            const Staged st = stage_entry(r_xy, r_co, r_cd);
            s_xy[threadIdx.x] = st.xy; s_ext[threadIdx.x] = st.ext;
            s_co[threadIdx.x] = st.co; s_cd[threadIdx.x] = st.cd;
            s_id[threadIdx.x] = r_id;
        } else {
            s_ext[threadIdx.x] = make_float2(-1.f, -1.f);
            s_xy[threadIdx.x] = make_float2(0.f, 0.f);
        }
        __syncthreads();
        {
            const int nxt = (r + 1) * GDR_BLOCK + (int)threadIdx.x;
            r_valid = nxt < total;
            if (r_valid) {
                r_id = point_list[range.y - 1 - nxt];
                r_xy = xy[r_id]; r_co = conic_opacity[r_id]; r_cd = rgbd[r_id];
            }
        }
        const int top = total - 1 - r * GDR_BLOCK;  // list position of LDS entry e: top - e
        if (top - (GDR_BLOCK - 1) >= wave_last) continue;
#pragma unroll 1
        for (int g = 0; g < GDR_BLOCK / GDR_WAVE; ++g) {
            const int gtop = top - g * GDR_WAVE;  // position of this group's entry 0
            if (gtop - (GDR_WAVE - 1) >= wave_last) continue;
            uint64_t m0, m1, m2, m3;
            {
                const float2 m = s_xy[g * GDR_WAVE + (int)lane];
                const float2 h = s_ext[g * GDR_WAVE + (int)lane];
                const bool v = h.x >= 0.f;
                const float lo_x = m.x - h.x, hi_x = m.x + h.x, lo_y = m.y - h.y, hi_y = m.y + h.y;
                const bool x0 = v && hi_x >= XA && lo_x <= XA + 3.f, x1 = v && hi_x >= XA + 4.f && lo_x <= XA + 7.f;
                const bool y0 = hi_y >= YA && lo_y <= YA + 3.f, y1 = hi_y >= YA + 4.f && lo_y <= YA + 7.f;
                const int mypos = gtop - (int)lane;
                m0 = __ballot(x0 && y0 && mypos < rl0);
                m1 = __ballot(x1 && y0 && mypos < rl1);
                m2 = __ballot(x0 && y1 && mypos < rl2);
                m3 = __ballot(x1 && y1 && mypos < rl3);
            }
            while ((m0 | m1 | m2 | m3) != 0ull) {
                const int e0 = pick_next(m0), e1 = pick_next(m1), e2 = pick_next(m2), e3 = pick_next(m3);
                const int es = row == 0 ? e0 : (row == 1 ? e1 : (row == 2 ? e2 : e3));
                const bool act = es < 64;
                const int e = g * GDR_WAVE + (act ? es : 0);
                const int pos = top - e;
                const float2 m = s_xy[e];
                const float4 co = s_co[e];
                const float dx = m.x - pxf, dy = m.y - pyf;
                const float p2 = gauss_power(dx, dy, co.x, co.y, co.z);
                const float G = __builtin_amdgcn_exp2f(p2);
                const float alpha = fminf(0.99f, co.w * G);
                const bool hit = act && (pos < last_contributor) && !(p2 > 0.f) && !(alpha < (1.f / 255.f));
                const uint64_t hb = __ballot(hit);
                if (hb == 0ull) continue;
                const float4 cd = s_cd[e];
                const float oma = 1.f - alpha;
                const float Tn = T / oma;
                const float w = hit ? alpha * Tn : 0.f;
                const float n0 = last_alpha * lc0 + (1.f - last_alpha) * acc0;
                const float n1 = last_alpha * lc1 + (1.f - last_alpha) * acc1;
                const float n2 = last_alpha * lc2 + (1.f - last_alpha) * acc2;
                const float nD = last_alpha * last_depth + (1.f - last_alpha) * accD;
                const float nA = last_alpha + (1.f - last_alpha) * accA;
                float dL_dalpha = (cd.x - n0) * gC0 + (cd.y - n1) * gC1 + (cd.z - n2) * gC2;
                dL_dalpha += (cd.w - nD) * gD;
                dL_dalpha += (1.f - nA) * gA;
                dL_dalpha *= Tn;
                dL_dalpha += (-T_final / oma) * bg_dot;
                dL_dalpha = hit ? dL_dalpha : 0.f;
                if (hit) {
                    T = Tn; acc0 = n0; acc1 = n1; acc2 = n2; accD = nD; accA = nA;
                    lc0 = cd.x; lc1 = cd.y; lc2 = cd.z; last_depth = cd.w; last_alpha = alpha;
                }
                const float dL_dG = co.w * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                float v_mx = dL_dG * (-gdx * co.x - gdy * co.y) * kx;
                float v_my = dL_dG * (-gdy * co.z - gdx * co.y) * ky;
                float v_ax = fabsf(v_mx), v_ay = fabsf(v_my);
                float v_cx = -0.5f * gdx * dx * dL_dG;
                float v_cy = -gdx * dy * dL_dG;
                float v_cz = -0.5f * gdy * dy * dL_dG;
                float v_dd = w * gD, v_r = w * gC0, v_g = w * gC1, v_b = w * gC2;
                float v_o = G * dL_dalpha;
                v_mx = row_sum(v_mx); v_my = row_sum(v_my); v_ax = row_sum(v_ax); v_ay = row_sum(v_ay);
                v_cx = row_sum(v_cx); v_cy = row_sum(v_cy); v_cz = row_sum(v_cz); v_dd = row_sum(v_dd);
                v_r = row_sum(v_r);   v_g = row_sum(v_g);   v_b = row_sum(v_b);   v_o = row_sum(v_o);
                // one lane per row that had a hit publishes the row totals
                if (li == 0 && ((hb >> (16 * row)) & 0xFFFFull) != 0ull) {
                    float* a = s_acc + e * GDR_ACC_STRIDE;
                    atomicAdd(a + 0, v_mx); atomicAdd(a + 1, v_my); atomicAdd(a + 2, v_ax); atomicAdd(a + 3, v_ay);
                    atomicAdd(a + 4, v_cx); atomicAdd(a + 5, v_cy); atomicAdd(a + 6, v_cz); atomicAdd(a + 7, v_dd);
                    atomicAdd(a + 8, v_r);  atomicAdd(a + 9, v_g);  atomicAdd(a + 10, v_b); atomicAdd(a + 11, v_o);
                    a[12] = 1.f;
                }
            }
        }
    }
}

}  // namespace

// GDR_RENDER_VARIANT=1|2 selects the older kernels (A/B measurements only); default 3
static int render_variant() {
    static const int v = [] { const char* e = getenv("GDR_RENDER_VARIANT"); return e ? atoi(e) : 3; }();
    return v;
}
static bool render_v1() { return render_variant() == 1; }

hipError_t launch_render_fwd(const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                             const gdr_image* img, const gdr_outputs* out, hipStream_t st) {
    const int W = s->image_width, H = s->image_height;
    const int gx = tile_grid_x(W), gy = tile_grid_y(H);
    const int ntiles = gx * gy;
    if (render_variant() >= 3)
        GDR_LAUNCH(GDR_K_RENDER_FWD, render_fwd_v3_kernel, dim3(ntiles), dim3(GDR_BLOCK), st,
                   (const uint2*)img->ranges, bin->values[bin->sorted], W, H, gx, ntiles,
                   (const float2*)g->xy, (const float4*)g->conic_opacity, (const float4*)g->rgb,
                   s->bg, img->final_T, img->n_contrib, out->color, out->depth, out->alpha);
    else if (!render_v1())
        GDR_LAUNCH(GDR_K_RENDER_FWD, render_fwd_v2_kernel, dim3(ntiles), dim3(GDR_BLOCK), st,
                   (const uint2*)img->ranges, bin->values[bin->sorted], W, H, gx, ntiles,
                   (const float2*)g->xy, (const float4*)g->conic_opacity, (const float4*)g->rgb,
                   s->bg, img->final_T, img->n_contrib, out->color, out->depth, out->alpha);
    else
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
    if (render_variant() >= 3)
        GDR_LAUNCH(GDR_K_RENDER_BWD, render_bwd_v3_kernel, dim3(ntiles), dim3(GDR_BLOCK), st,
                   (const uint2*)img->ranges, bin->values[bin->sorted], W, H, gx, ntiles, s->bg,
                   (const float2*)g->xy, (const float4*)g->conic_opacity, (const float4*)g->rgb,
                   img->final_T, img->n_contrib, gi->dL_dcolor, gi->dL_ddepth, gi->dL_dalpha,
                   go->dL_dmeans2D, go->scratch, go->dL_dopacities);
    else if (!render_v1())
        GDR_LAUNCH(GDR_K_RENDER_BWD, render_bwd_v2_kernel, dim3(ntiles), dim3(GDR_BLOCK), st,
                   (const uint2*)img->ranges, bin->values[bin->sorted], W, H, gx, ntiles, s->bg,
                   (const float2*)g->xy, (const float4*)g->conic_opacity, (const float4*)g->rgb,
                   img->final_T, img->n_contrib, gi->dL_dcolor, gi->dL_ddepth, gi->dL_dalpha,
                   go->dL_dmeans2D, go->scratch, go->dL_dopacities);
    else
    GDR_LAUNCH(GDR_K_RENDER_BWD, render_bwd_kernel, dim3(ntiles), dim3(GDR_BLOCK), st,
                       (const uint2*)img->ranges, bin->values[bin->sorted], W, H, gx, ntiles, s->bg,
                       (const float2*)g->xy, (const float4*)g->conic_opacity, (const float4*)g->rgb,
                       img->final_T, img->n_contrib, gi->dL_dcolor, gi->dL_ddepth, gi->dL_dalpha,
                       go->dL_dmeans2D, go->scratch, go->dL_dopacities);
    return hipGetLastError();
}

}  // namespace gdr
