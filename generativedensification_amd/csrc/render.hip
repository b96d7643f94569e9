// render.hip — K6 (per-tile alpha-composited forward) and K7 (per-pixel reverse-order
// backward) of the rasterizer for gfx950.
// Behaviour: SURVEY.md Appendix A.3 / A.4 (3DGS tile renderer + depth / alpha outputs
// + AbsGS |.|-accumulated screen-space gradients), i.e. what the reference obtains from
// rasterizer(...) at /root/reference/lightning/renderer.py:250-259 and differentiates at
// /root/reference/lightning/network.py:867-878.
//
// CDNA4 mapping (not the 32-wide warp layout of the CUDA lineage):
//   * one workgroup = one 16x16 tile = 4 wavefronts; a wave owns an 8x8 sub-tile and each
//     of its four 16-lane DPP rows composites its OWN 4x4 pixel block;
//   * the tile's Gaussian slice is staged 256 entries at a time in LDS (xy, alpha-extent,
//     conic*log2e + opacity, rgb + depth = 48 B/entry) while the next slice is already being
//     fetched into registers; every lane tests ONE staged entry against the four blocks and
//     four 64-bit ballots give each block its culled sub-list (masks live in SGPRs);
//   * the inner loop is branch-free/predicated; G = v_exp_f32 on a conic pre-scaled by
//     log2(e); early-out is per block (row) and per wave via ballots;
//   * backward: the 12 per-Gaussian partial gradients are reduce-scattered inside each
//     16-lane row with DPP (45 VALU ops, one total per lane) and published with ONE
//     global_atomic_add_f32 instruction into a 64-byte per-Gaussian record; |.| of the
//     mean2D terms is taken per pixel BEFORE the reduction (AbsGS semantics);
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

// Row-local reduce-scatter of 12 per-lane values over the 16 lanes of a DPP row: lane i of the
// row returns the row total of value i (i < 12; lanes 12..15 return 0).  Four halving
// exchanges (row_mirror, row_half_mirror, quad reverse, quad xor-1): 45 VALU ops instead of
// 12 x 4 full reductions, and the totals land one per lane, so ONE atomic instruction
// publishes all of them.
__device__ __forceinline__ float row_reduce_scatter12(const float (&v)[12], uint32_t li) {
    const bool b3 = li & 8u, b2 = li & 4u, b1 = li & 2u, b0 = li & 1u;
    float u[8], t[4], s2[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float hi = (j + 8 < 12) ? v[j + 8] : 0.f;
        const float keep = b3 ? hi : v[j], send = b3 ? v[j] : hi;
        u[j] = keep + dpp_get<0x140, 0xf>(send);  // row_mirror: partner 15 - i
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float keep = b2 ? u[j + 4] : u[j], send = b2 ? u[j] : u[j + 4];
        t[j] = keep + dpp_get<0x141, 0xf>(send);  // row_half_mirror: partner i ^ 7
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float keep = b1 ? t[j + 2] : t[j], send = b1 ? t[j] : t[j + 2];
        s2[j] = keep + dpp_get<0x1B, 0xf>(send);  // quad_perm [3,2,1,0]: partner i ^ 3
    }
    const float keep = b0 ? s2[1] : s2[0], send = b0 ? s2[0] : s2[1];
    return keep + dpp_get<0xB1, 0xf>(send);       // quad_perm [1,0,3,2]: partner i ^ 1
}

#define GDR_ROW_MASK(k) (0xFFFFull << (16 * (k)))

__global__ __launch_bounds__(GDR_BLOCK) void render_fwd_kernel(
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

__global__ __launch_bounds__(GDR_BLOCK) void render_bwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int gx,
    int ntiles, const float* __restrict__ bg, const float2* __restrict__ xy,
    const float4* __restrict__ conic_opacity, const float4* __restrict__ rgbd,
    const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dpix, const float* __restrict__ dL_ddepthpix,
    const float* __restrict__ dL_dalphapix, float* __restrict__ grad_rec) {
    __shared__ float2 s_xy[GDR_BLOCK];
    __shared__ float2 s_ext[GDR_BLOCK];
    __shared__ float4 s_co[GDR_BLOCK];
    __shared__ float4 s_cd[GDR_BLOCK];
    __shared__ uint32_t s_id[GDR_BLOCK];

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


    float2 r_xy = make_float2(0.f, 0.f);
    float4 r_co = make_float4(0.f, 0.f, 0.f, 0.f), r_cd = r_co;
    uint32_t r_id = 0;
    bool r_valid = (int)threadIdx.x < total;
    if (r_valid) {
        r_id = point_list[range.y - 1 - threadIdx.x];
        r_xy = xy[r_id]; r_co = conic_opacity[r_id]; r_cd = rgbd[r_id];
    }
    for (int r = 0; r < rounds; ++r) {
        __syncthreads();  // every wave finished reading the previous slice
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
                const float vals[12] = {v_mx, v_my, v_ax, v_ay, v_cx, v_cy, v_cz, v_dd, v_r, v_g, v_b, v_o};
                const float tot = row_reduce_scatter12(vals, li);
                // lanes 0..12 of every row that had a hit publish the row totals: one DS instruction
                // lanes 0..11 of every row that had a hit add the row totals to the Gaussian's
                // 64-byte gradient record: one global_atomic_add_f32 instruction, one cache line
                // per row (no return value => fire and forget)
                if (li < 12u && ((hb >> (16 * row)) & 0xFFFFull) != 0ull)
                    atomicAdd(grad_rec + 16 * (size_t)s_id[e] + li, tot);
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
               (const float2*)g->xy, (const float4*)g->conic_opacity, (const float4*)g->rgb, s->bg,
               img->final_T, img->n_contrib, out->color, out->depth, out->alpha);
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
               go->scratch);
    return hipGetLastError();
}

}  // namespace gdr
