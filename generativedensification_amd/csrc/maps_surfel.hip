// maps_surfel.hip — the per-pixel maps the 2DGS adaptor derives from the rasterizer's allmap, fused
// (/root/reference/lightning/renderer_2dgs.py:241-278, SURVEY §8f-3 "depth_to_normal fused"):
//   acc_map      = allmap[1]
//   rend_normal  = allmap[2:5] rotated to world space by world_view[:3,:3]^T                     (:244-246)
//   surf_depth   = (1 - r) nan0(allmap[0] / alpha) + r nan0(allmap[5])                           (:248-262)
//   depth_normal = normalize(cross(P(y+1,x) - P(y-1,x), P(y,x+1) - P(y,x-1))) alpha (alpha detached, zero on the
//                  1-pixel border), P = ray origin + surf_depth * ray direction                   (:75-90, 266-271)
//   rend_dist    = allmap[6]
// In torch this is ~45 elementwise / slicing / small-GEMM launches forward and ~90 backward per view over 640k
// pixels; here one kernel forward and two backward (the stencil's transpose needs the neighbours' partials: pass A
// stores dL/d(row difference), dL/d(column difference) per pixel, pass B gathers them).  HBM-bound, ~150 B/pixel.
#include "gdr_common.h"

namespace gdr {
namespace {

// torch.nan_to_num(x, 0, 0) of renderer_2dgs.py:249,254: nan -> 0, +inf -> 0, and -inf -> the most negative finite
// float (neginf is left at its default there); none of the three passes a gradient
#define GSR_NEG_MAX (-3.402823466e+38f)
__device__ __forceinline__ float nan0(float v) { return (v != v || v == INFINITY) ? 0.f : (v == -INFINITY ? GSR_NEG_MAX : v); }

struct Maps {
    const float* allmap; const float* rays; int H, W; float r;
};

// surf_depth of pixel q and whether its expected-depth quotient is differentiable there
__device__ __forceinline__ float surf_depth(const Maps& m, int q, size_t P, bool* exp_ok, bool* med_ok) {
    const float D = m.allmap[q], a = m.allmap[P + q], med = m.allmap[5 * P + q];
    const float e = D / a;
    // differentiable only where the value is finite (nan_to_num's backward masks nan and both infinities)
    const bool eo = !(e != e) && e != INFINITY && e != -INFINITY, mo = !(med != med) && med != INFINITY && med != -INFINITY;
    if (exp_ok) *exp_ok = eo;
    if (med_ok) *med_ok = mo;
    return (1.f - m.r) * nan0(e) + m.r * nan0(med);
}

__device__ __forceinline__ void point_at(const Maps& m, int q, size_t P, float* p) {
    const float sd = surf_depth(m, q, P, nullptr, nullptr);
    const float* ry = m.rays + 6 * (size_t)q;
    p[0] = fmaf(sd, ry[3], ry[0]); p[1] = fmaf(sd, ry[4], ry[1]); p[2] = fmaf(sd, ry[5], ry[2]);
}

struct Stencil { float a[3], b[3], c[3], len; };  // a = row difference, b = column difference, c = a x b

__device__ __forceinline__ void stencil_at(const Maps& m, int y, int x, size_t P, Stencil& s) {
    float u[3], d[3], l[3], r[3];
    point_at(m, (y - 1) * m.W + x, P, u); point_at(m, (y + 1) * m.W + x, P, d);
    point_at(m, y * m.W + x - 1, P, l); point_at(m, y * m.W + x + 1, P, r);
#pragma unroll
    for (int k = 0; k < 3; ++k) { s.a[k] = d[k] - u[k]; s.b[k] = r[k] - l[k]; }
    s.c[0] = s.a[1] * s.b[2] - s.a[2] * s.b[1];
    s.c[1] = s.a[2] * s.b[0] - s.a[0] * s.b[2];
    s.c[2] = s.a[0] * s.b[1] - s.a[1] * s.b[0];
    s.len = sqrtf(fmaf(s.c[0], s.c[0], fmaf(s.c[1], s.c[1], s.c[2] * s.c[2])));
}

__global__ __launch_bounds__(GDR_BLOCK) void surfel_maps_fwd_kernel(Maps m, const float* __restrict__ view,
                                                                     float* __restrict__ depth, float* __restrict__ acc,
                                                                     float* __restrict__ rend_normal,
                                                                     float* __restrict__ depth_normal,
                                                                     float* __restrict__ rend_dist) {
    const int q = blockIdx.x * GDR_BLOCK + threadIdx.x;
    const size_t P = (size_t)m.H * m.W;
    if (q >= (int)P) return;
    const int y = q / m.W, x = q - y * m.W;
    const float a = m.allmap[P + q];
    depth[q] = surf_depth(m, q, P, nullptr, nullptr);
    acc[q] = a;
    rend_dist[q] = m.allmap[6 * P + q];
    const float n0 = m.allmap[2 * P + q], n1 = m.allmap[3 * P + q], n2 = m.allmap[4 * P + q];
#pragma unroll
    for (int j = 0; j < 3; ++j) rend_normal[3 * (size_t)q + j] = fmaf(view[4 * j], n0, fmaf(view[4 * j + 1], n1, view[4 * j + 2] * n2));
    float dn[3] = {0.f, 0.f, 0.f};
    if (y >= 1 && y < m.H - 1 && x >= 1 && x < m.W - 1) {
        Stencil s;
        stencil_at(m, y, x, P, s);
        const float inv = a / fmaxf(s.len, 1e-12f);  // F.normalize(eps = 1e-12), times the (detached) alpha
        dn[0] = s.c[0] * inv; dn[1] = s.c[1] * inv; dn[2] = s.c[2] * inv;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) depth_normal[3 * (size_t)q + j] = dn[j];
}

// pass A: per interior pixel, dL/d(row difference) and dL/d(column difference) of its normal -> scratch (6 floats)
__global__ __launch_bounds__(GDR_BLOCK) void surfel_maps_bwd_a_kernel(Maps m, const float* __restrict__ g_dn,
                                                                       float* __restrict__ scratch) {
    const int q = blockIdx.x * GDR_BLOCK + threadIdx.x;
    const size_t P = (size_t)m.H * m.W;
    if (q >= (int)P) return;
    const int y = q / m.W, x = q - y * m.W;
    float ga[3] = {0.f, 0.f, 0.f}, gb[3] = {0.f, 0.f, 0.f};
    if (g_dn && y >= 1 && y < m.H - 1 && x >= 1 && x < m.W - 1) {
        Stencil s;
        stencil_at(m, y, x, P, s);
        const float a = m.allmap[P + q];
        const float g[3] = {g_dn[3 * (size_t)q] * a, g_dn[3 * (size_t)q + 1] * a, g_dn[3 * (size_t)q + 2] * a};
        float gc[3];
        if (s.len > 1e-12f) {  // n = c / |c|: dL/dc = (g - n (n.g)) / |c|
            const float inv = 1.f / s.len;
            const float n[3] = {s.c[0] * inv, s.c[1] * inv, s.c[2] * inv};
            const float dot = fmaf(n[0], g[0], fmaf(n[1], g[1], n[2] * g[2]));
#pragma unroll
            for (int k = 0; k < 3; ++k) gc[k] = (g[k] - n[k] * dot) * inv;
        } else {               // clamped denominator: n = c / eps
#pragma unroll
            for (int k = 0; k < 3; ++k) gc[k] = g[k] * 1e12f;
        }
        // c = a x b: dL/da = b x gc, dL/db = gc x a
        ga[0] = s.b[1] * gc[2] - s.b[2] * gc[1]; ga[1] = s.b[2] * gc[0] - s.b[0] * gc[2]; ga[2] = s.b[0] * gc[1] - s.b[1] * gc[0];
        gb[0] = gc[1] * s.a[2] - gc[2] * s.a[1]; gb[1] = gc[2] * s.a[0] - gc[0] * s.a[2]; gb[2] = gc[0] * s.a[1] - gc[1] * s.a[0];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { scratch[(size_t)k * P + q] = ga[k]; scratch[(size_t)(3 + k) * P + q] = gb[k]; }
}

// pass B: gather the neighbours' partials into dL/dsurf_depth, then everything into dL/dallmap
__global__ __launch_bounds__(GDR_BLOCK) void surfel_maps_bwd_b_kernel(Maps m, const float* __restrict__ view,
                                                                       const float* __restrict__ g_depth,
                                                                       const float* __restrict__ g_acc,
                                                                       const float* __restrict__ g_rn,
                                                                       const float* __restrict__ g_dist,
                                                                       const float* __restrict__ scratch, int have_dn,
                                                                       float* __restrict__ dL_dallmap) {
    const int q = blockIdx.x * GDR_BLOCK + threadIdx.x;
    const size_t P = (size_t)m.H * m.W;
    if (q >= (int)P) return;
    const int y = q / m.W, x = q - y * m.W;
    float gp[3] = {0.f, 0.f, 0.f};
    if (have_dn) {
        const bool xin = x >= 1 && x < m.W - 1, yin = y >= 1 && y < m.H - 1;
        // q is the lower neighbour of (y-1,x) [+ row term], the upper one of (y+1,x) [-], the right one of (y,x-1)
        // [+ column term], the left one of (y,x+1) [-]; the source pixel must be interior
        if (xin && y - 1 >= 1 && y - 1 < m.H - 1)
#pragma unroll
            for (int k = 0; k < 3; ++k) gp[k] += scratch[(size_t)k * P + q - m.W];
        if (xin && y + 1 >= 1 && y + 1 < m.H - 1)
#pragma unroll
            for (int k = 0; k < 3; ++k) gp[k] -= scratch[(size_t)k * P + q + m.W];
        if (yin && x - 1 >= 1 && x - 1 < m.W - 1)
#pragma unroll
            for (int k = 0; k < 3; ++k) gp[k] += scratch[(size_t)(3 + k) * P + q - 1];
        if (yin && x + 1 >= 1 && x + 1 < m.W - 1)
#pragma unroll
            for (int k = 0; k < 3; ++k) gp[k] -= scratch[(size_t)(3 + k) * P + q + 1];
    }
    const float* ry = m.rays + 6 * (size_t)q;
    float gsd = fmaf(gp[0], ry[3], fmaf(gp[1], ry[4], gp[2] * ry[5]));
    if (g_depth) gsd += g_depth[q];
    bool eo, mo;
    surf_depth(m, q, P, &eo, &mo);
    const float D = m.allmap[q], a = m.allmap[P + q];
    const float ge = eo ? (1.f - m.r) * gsd : 0.f;
    // an empty pixel (alpha = 0) has no dependence on D or alpha (nan_to_num); torch would hand 0/0 = NaN here
    const float gD = (eo && a != 0.f) ? ge / a : 0.f;
    float gA = (eo && a != 0.f) ? -ge * D / (a * a) : 0.f;
    if (g_acc) gA += g_acc[q];
    dL_dallmap[q] = gD;
    dL_dallmap[P + q] = gA;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float v = 0.f;
        if (g_rn) v = fmaf(view[i], g_rn[3 * (size_t)q], fmaf(view[4 + i], g_rn[3 * (size_t)q + 1], view[8 + i] * g_rn[3 * (size_t)q + 2]));
        dL_dallmap[(2 + i) * P + q] = v;
    }
    dL_dallmap[5 * P + q] = mo ? m.r * gsd : 0.f;
    dL_dallmap[6 * P + q] = g_dist ? g_dist[q] : 0.f;
}

// ---------------------------------------------------------------------------------
// Fused measurement / training loss of the surfel path on top of the same maps:
//   L = mean_{c,p}(clamp(color,0,1) - target)^2                       (renderer_2dgs.py:236, loss.py:37-38)
//     + w_dist mean(rend_dist) + w_normal mean((1 - <rend_normal, depth_normal>) acc.detach())   (loss.py:49-61)
//     + w_depth mean(surf_depth) + w_alpha mean(acc)                  (coverage of the depth / alpha paths, §8d)
// forward: one reduction kernel; backward: the two stencil passes with the upstream gradients of the five maps
// written out analytically (never materialised) + dL/dcolor.
// ---------------------------------------------------------------------------------
struct LossW { float dist, normal, depth, alpha; };

__global__ __launch_bounds__(GDR_BLOCK) void surfel_loss_fwd_kernel(Maps m, const float* __restrict__ view,
                                                                     const float* __restrict__ color,
                                                                     const float* __restrict__ target, LossW w,
                                                                     float* __restrict__ loss) {
    __shared__ float wsum[GDR_BLOCK / GDR_WAVE];
    const size_t P = (size_t)m.H * m.W;
    const float invp = 1.f / (float)P;
    float acc = 0.f;
    for (int q = blockIdx.x * GDR_BLOCK + threadIdx.x; q < (int)P; q += gridDim.x * GDR_BLOCK) {
        const int y = q / m.W, x = q - y * m.W;
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d = fminf(fmaxf(color[(size_t)c * P + q], 0.f), 1.f) - target[(size_t)c * P + q];
            se = fmaf(d, d, se);
        }
        const float a = m.allmap[P + q];
        float dot = 0.f;
        if (y >= 1 && y < m.H - 1 && x >= 1 && x < m.W - 1) {
            Stencil s;
            stencil_at(m, y, x, P, s);
            const float inv = a / fmaxf(s.len, 1e-12f);
            const float n0 = m.allmap[2 * P + q], n1 = m.allmap[3 * P + q], n2 = m.allmap[4 * P + q];
#pragma unroll
            for (int j = 0; j < 3; ++j)
                dot = fmaf(fmaf(view[4 * j], n0, fmaf(view[4 * j + 1], n1, view[4 * j + 2] * n2)), s.c[j] * inv, dot);
        }
        acc += (se * (1.f / 3.f) + w.dist * m.allmap[6 * P + q] + w.normal * (1.f - dot) * a +
                w.depth * surf_depth(m, q, P, nullptr, nullptr) + w.alpha * a) * invp;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
}

// pass A: as surfel_maps_bwd_a with g_depth_normal = -(w_normal/P) acc rend_normal go; also keeps the pixel's
// depth_normal (3 more floats) for pass B's rend_normal gradient
__global__ __launch_bounds__(GDR_BLOCK) void surfel_loss_bwd_a_kernel(Maps m, const float* __restrict__ view, LossW w,
                                                                       const float* __restrict__ go_ptr,
                                                                       float* __restrict__ scratch) {
    const int q = blockIdx.x * GDR_BLOCK + threadIdx.x;
    const size_t P = (size_t)m.H * m.W;
    if (q >= (int)P) return;
    const int y = q / m.W, x = q - y * m.W;
    const float go = go_ptr ? *go_ptr : 1.f;
    float ga[3] = {0.f, 0.f, 0.f}, gb[3] = {0.f, 0.f, 0.f}, dn[3] = {0.f, 0.f, 0.f};
    if (y >= 1 && y < m.H - 1 && x >= 1 && x < m.W - 1) {
        Stencil s;
        stencil_at(m, y, x, P, s);
        const float a = m.allmap[P + q];
        const float n0 = m.allmap[2 * P + q], n1 = m.allmap[3 * P + q], n2 = m.allmap[4 * P + q];
        const float k = -(w.normal / (float)P) * a * a * go;  // upstream of the UNIT normal: g_dn * alpha
        float g[3], gc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) g[j] = k * fmaf(view[4 * j], n0, fmaf(view[4 * j + 1], n1, view[4 * j + 2] * n2));
        const float invl = 1.f / fmaxf(s.len, 1e-12f);
#pragma unroll
        for (int j = 0; j < 3; ++j) dn[j] = s.c[j] * invl * a;
        if (s.len > 1e-12f) {
            const float nn[3] = {s.c[0] * invl, s.c[1] * invl, s.c[2] * invl};
            const float dot = fmaf(nn[0], g[0], fmaf(nn[1], g[1], nn[2] * g[2]));
#pragma unroll
            for (int j = 0; j < 3; ++j) gc[j] = (g[j] - nn[j] * dot) * invl;
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) gc[j] = g[j] * 1e12f;
        }
        ga[0] = s.b[1] * gc[2] - s.b[2] * gc[1]; ga[1] = s.b[2] * gc[0] - s.b[0] * gc[2]; ga[2] = s.b[0] * gc[1] - s.b[1] * gc[0];
        gb[0] = gc[1] * s.a[2] - gc[2] * s.a[1]; gb[1] = gc[2] * s.a[0] - gc[0] * s.a[2]; gb[2] = gc[0] * s.a[1] - gc[1] * s.a[0];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        scratch[(size_t)k * P + q] = ga[k]; scratch[(size_t)(3 + k) * P + q] = gb[k]; scratch[(size_t)(6 + k) * P + q] = dn[k];
    }
}

__global__ __launch_bounds__(GDR_BLOCK) void surfel_loss_bwd_b_kernel(Maps m, const float* __restrict__ view,
                                                                       const float* __restrict__ color,
                                                                       const float* __restrict__ target, LossW w,
                                                                       const float* __restrict__ go_ptr,
                                                                       const float* __restrict__ scratch,
                                                                       float* __restrict__ dL_dcolor,
                                                                       float* __restrict__ dL_dallmap) {
    const int q = blockIdx.x * GDR_BLOCK + threadIdx.x;
    const size_t P = (size_t)m.H * m.W;
    if (q >= (int)P) return;
    const int y = q / m.W, x = q - y * m.W;
    const float go = go_ptr ? *go_ptr : 1.f;
    const float invp = go / (float)P;
    const float kc = 2.f * invp / 3.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = color[(size_t)c * P + q];
        const float d = fminf(fmaxf(v, 0.f), 1.f) - target[(size_t)c * P + q];
        dL_dcolor[(size_t)c * P + q] = (v >= 0.f && v <= 1.f) ? kc * d : 0.f;
    }
    float gp[3] = {0.f, 0.f, 0.f};
    const bool xin = x >= 1 && x < m.W - 1, yin = y >= 1 && y < m.H - 1;
    if (xin && y - 1 >= 1 && y - 1 < m.H - 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) gp[k] += scratch[(size_t)k * P + q - m.W];
    if (xin && y + 1 >= 1 && y + 1 < m.H - 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) gp[k] -= scratch[(size_t)k * P + q + m.W];
    if (yin && x - 1 >= 1 && x - 1 < m.W - 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) gp[k] += scratch[(size_t)(3 + k) * P + q - 1];
    if (yin && x + 1 >= 1 && x + 1 < m.W - 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) gp[k] -= scratch[(size_t)(3 + k) * P + q + 1];
    const float* ry = m.rays + 6 * (size_t)q;
    const float gsd = fmaf(gp[0], ry[3], fmaf(gp[1], ry[4], gp[2] * ry[5])) + w.depth * invp;
    bool eo, mo;
    surf_depth(m, q, P, &eo, &mo);
    const float D = m.allmap[q], a = m.allmap[P + q];
    const float ge = eo ? (1.f - m.r) * gsd : 0.f;
    dL_dallmap[q] = (eo && a != 0.f) ? ge / a : 0.f;
    dL_dallmap[P + q] = ((eo && a != 0.f) ? -ge * D / (a * a) : 0.f) + w.alpha * invp;  // acc is detached in the normal term
    const float kn = -w.normal * invp * a;  // d/d rend_normal = -(w_normal/P) acc depth_normal
    const float d0 = scratch[(size_t)6 * P + q], d1 = scratch[(size_t)7 * P + q], d2 = scratch[(size_t)8 * P + q];
#pragma unroll
    for (int i = 0; i < 3; ++i) dL_dallmap[(2 + i) * P + q] = kn * fmaf(view[i], d0, fmaf(view[4 + i], d1, view[8 + i] * d2));
    dL_dallmap[5 * P + q] = mo ? m.r * gsd : 0.f;
    dL_dallmap[6 * P + q] = w.dist * invp;
}

}  // namespace

hipError_t launch_surfel_loss_fwd(const float* color, const float* allmap, const float* rays, const float* view,
                                  const float* target, int H, int W, float r, float w_dist, float w_normal, float w_depth,
                                  float w_alpha, float* loss, hipStream_t st) {
    const Maps m{allmap, rays, H, W, r};
    const LossW w{w_dist, w_normal, w_depth, w_alpha};
    const int grid = min(div_up((int64_t)H * W, GDR_BLOCK), 2048);
    GDR_LAUNCH(GDR_K_VIEW_LOSS, surfel_loss_fwd_kernel, dim3(grid), dim3(GDR_BLOCK), st, m, view, color, target, w, loss);
    return hipGetLastError();
}

hipError_t launch_surfel_loss_bwd(const float* color, const float* allmap, const float* rays, const float* view,
                                  const float* target, int H, int W, float r, float w_dist, float w_normal, float w_depth,
                                  float w_alpha, const float* g, float* scratch, float* dL_dcolor, float* dL_dallmap,
                                  hipStream_t st) {
    const Maps m{allmap, rays, H, W, r};
    const LossW w{w_dist, w_normal, w_depth, w_alpha};
    const dim3 grid(div_up((int64_t)H * W, GDR_BLOCK));
    GDR_LAUNCH(GDR_K_VIEW_LOSS, surfel_loss_bwd_a_kernel, grid, dim3(GDR_BLOCK), st, m, view, w, g, scratch);
    GDR_LAUNCH(GDR_K_VIEW_LOSS, surfel_loss_bwd_b_kernel, grid, dim3(GDR_BLOCK), st, m, view, color, target, w, g, scratch,
               dL_dcolor, dL_dallmap);
    return hipGetLastError();
}

hipError_t launch_surfel_maps_fwd(const float* allmap, const float* rays, const float* view, int H, int W, float r,
                                  float* depth, float* acc, float* rend_normal, float* depth_normal, float* rend_dist,
                                  hipStream_t st) {
    const Maps m{allmap, rays, H, W, r};
    GDR_LAUNCH(GDR_K_SURFEL_MAPS, surfel_maps_fwd_kernel, dim3(div_up((int64_t)H * W, GDR_BLOCK)), dim3(GDR_BLOCK), st, m,
               view, depth, acc, rend_normal, depth_normal, rend_dist);
    return hipGetLastError();
}

hipError_t launch_surfel_maps_bwd(const float* allmap, const float* rays, const float* view, int H, int W, float r,
                                  const float* g_depth, const float* g_acc, const float* g_rn, const float* g_dn,
                                  const float* g_dist, float* scratch, float* dL_dallmap, hipStream_t st) {
    const Maps m{allmap, rays, H, W, r};
    const dim3 grid(div_up((int64_t)H * W, GDR_BLOCK));
    if (g_dn) GDR_LAUNCH(GDR_K_SURFEL_MAPS, surfel_maps_bwd_a_kernel, grid, dim3(GDR_BLOCK), st, m, g_dn, scratch);
    GDR_LAUNCH(GDR_K_SURFEL_MAPS, surfel_maps_bwd_b_kernel, grid, dim3(GDR_BLOCK), st, m, view, g_depth, g_acc, g_rn,
               g_dist, scratch, g_dn ? 1 : 0, dL_dallmap);
    return hipGetLastError();
}

}  // namespace gdr
