// render_surfel.hip — K6s (per-tile alpha-composited forward) and K7s (per-pixel reverse-order backward) of the
// 2D-Gaussian (surfel) rasterizer for gfx950 (include/gsr.h): what the reference obtains from
// `diff_surfel_rasterization` at /root/reference/lightning/renderer_2dgs.py:224-234 (image + the 7-channel allmap
// sliced at :241-257); arithmetic as restated in oracle/gsr_oracle.c.
//
// Same CDNA4 mapping as render.hip (measured there): workgroup = 16x16 tile, wave = 8x8 sub-tile,
// each 16-lane DPP row composites its OWN 4x4 pixel block against its OWN sub-list — every lane tests one staged
// surfel's conservative {alpha >= 1/255} box (written by K1s) against the wave's four blocks and four ballots give
// the per-block masks; idle rows read a null LDS entry (T = 0 -> no intersection); predication by float
// thresholds; the tile's sorted slice is staged 256 entries at a time while the next one is in flight.
// Per pixel and surfel: ray-splat intersection (u,v) = cross(x Tw - Tu, y Tw - Tv) dehomogenised (v_rcp_f32),
// G = v_exp_f32(-0.5 log2e min(u^2+v^2, 2|pix - centre|^2)).  Backward: all "behind" recurrences (colour, depth,
// coverage, normal and the distortion weight) share the select-free form B <- B + a (c - B); the 20 per-surfel
// partials are reduce-scattered inside each row (16 + 4 values) and published with two atomic instructions into a
// 128-byte gradient record; K7s walks slice-wide row lists (render_common.h GDR_REFILL): 647 -> 540 us at C5 (a
// lockstep variant — all four rows on the union list, rows summed with v_permlane16/32_swap, one atomic per wave —
// reached 578 us and was dropped for it).
#include <stdlib.h>

#include <string.h>

#include "gdr_common.h"
#include "render_common.h"
#include "../../include/gsr.h"

namespace gdr {

namespace {

#define GSR_NEAR 0.2f
#define GSR_FAR 100.0f
#define GSR_FARK (GSR_FAR / (GSR_FAR - GSR_NEAR))

struct SEntry {  // one staged list entry, in registers
    uint32_t e;
    float4 tu, tv, tw, nr;  // (Tu, cx) (Tv, cy) (Tw, opacity) (normal, r)
    float2 gb;
};

struct SurfelLds {
    float4 tu[GDR_BLOCK + 1], tv[GDR_BLOCK + 1], tw[GDR_BLOCK + 1], nr[GDR_BLOCK + 1];
    float2 gb[GDR_BLOCK + 1];
    float4 box[GDR_BLOCK];  // lo.x, lo.y, hi.x, hi.y
};

// the 96-byte render record of one surfel is prefetched into six float4 registers (q0..q5) and staged from there
#define GSR_LOAD_REC(ID)                                                                   \
    do {                                                                                   \
        const float4* p_ = rec + 6 * (size_t)(ID);                                         \
        q0 = p_[0]; q1 = p_[1]; q2 = p_[2]; q3 = p_[3]; q4 = p_[4]; q5 = p_[5];            \
    } while (0)
#define GSR_STAGE(VALID)                                                                   \
    do {                                                                                   \
        if (VALID) {                                                                       \
            lds.tu[threadIdx.x] = q0; lds.tv[threadIdx.x] = q1; lds.tw[threadIdx.x] = q2;  \
            lds.nr[threadIdx.x] = q3; lds.gb[threadIdx.x] = make_float2(q4.x, q4.y);       \
            lds.box[threadIdx.x] = make_float4(q4.z, q4.w, q5.x, q5.y);                    \
        } else { /* culled for every block */                                              \
            lds.box[threadIdx.x] = make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);  \
        }                                                                                  \
    } while (0)

__device__ __forceinline__ void null_entry(SurfelLds& s) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    s.tu[GDR_NULL_ENTRY] = z; s.tv[GDR_NULL_ENTRY] = z; s.tw[GDR_NULL_ENTRY] = z; s.nr[GDR_NULL_ENTRY] = z;
    s.gb[GDR_NULL_ENTRY] = make_float2(0.f, 0.f);
}

__device__ __forceinline__ void block_masks(const SurfelLds& s, int g, float XA, float YA, bool a0, bool a1, bool a2,
                                            bool a3, uint64_t& m0, uint64_t& m1, uint64_t& m2, uint64_t& m3, bool (&mine)[4]) {
    const float4 b = s.box[g * GDR_WAVE + (int)lane_id()];
    const bool x0 = b.z >= XA && b.x <= XA + 3.f, x1 = b.z >= XA + 4.f && b.x <= XA + 7.f;
    const bool y0 = b.w >= YA && b.y <= YA + 3.f, y1 = b.w >= YA + 4.f && b.y <= YA + 7.f;
    mine[0] = x0 && y0 && a0; mine[1] = x1 && y0 && a1; mine[2] = x0 && y1 && a2; mine[3] = x1 && y1 && a3;
    m0 = __ballot(mine[0]);
    m1 = __ballot(mine[1]);
    m2 = __ballot(mine[2]);
    m3 = __ballot(mine[3]);
}

// ray-splat intersection and Gaussian weight of one pixel against one staged surfel
struct Hit {
    float kx, ky, kz, lx, ly, lz, rz, sx, sy, dx, dy, depth, G, alpha;
    bool use3d, geom_ok;
};

// The ray-splat intersection of one pixel against one staged surfel (oracle/gsr_oracle.c surfel_eval).
// The published formulation is ill-conditioned in fp32: k = x Tw - Tu cancels (~800 * 2 against ~1600 for a small surfel
// far from the image origin), the cross product cancels again, and (u, v) = (cx, cy) / cz divides by what is left — two
// fp32 evaluations that differ in ONE rounding (an fma instead of multiply + subtract, a reciprocal instead of a
// division) differ in their single worst gradient element by as much as either differs from float64, in either direction
// (round 3: HIP's worst element of a C5 backward sat 2.5x further from float64 than the f32 oracle's in one test and 3x
// closer in the next; round 4 measured that making the reciprocal and the exponent MORE accurate than the oracle's moves
// nothing: profiles/r04_surfel_numerics.txt).  GSR_ORACLE_ORDER (default): the geometry is evaluated in the oracle's own
// operation order — multiply then subtract for k, l and the cross product (no contraction), a correctly rounded quotient
// (hardware reciprocal + one Newton step + one residual correction per quotient: 6 fma for both instead of two ~10-
// instruction IEEE divisions), products and sums of rho and depth unfused — so (u, v), rho and the depth of every
// (pixel, surfel) pair are the f32 oracle's BIT FOR BIT and the two programs differ only where they are well conditioned
// (exp: <= 2 ulp; the order of fp32 sums).  -DGSR_ORACLE_ORDER=0 = the round-3 arithmetic (A/B builds only).
#ifndef GSR_ORACLE_ORDER
#define GSR_ORACLE_ORDER 1
#endif
__device__ __forceinline__ void intersect(const SEntry& en, float pxf, float pyf, Hit& h) {
#if GSR_ORACLE_ORDER
#pragma clang fp contract(off)
    h.kx = pxf * en.tw.x - en.tu.x; h.ky = pxf * en.tw.y - en.tu.y; h.kz = pxf * en.tw.z - en.tu.z;
    h.lx = pyf * en.tw.x - en.tv.x; h.ly = pyf * en.tw.y - en.tv.y; h.lz = pyf * en.tw.z - en.tv.z;
    const float cx = h.ky * h.lz - h.kz * h.ly, cy = h.kz * h.lx - h.kx * h.lz, cz = h.kx * h.ly - h.ky * h.lx;
    {
        const float r0 = __builtin_amdgcn_rcpf(cz);
        const float r = __builtin_fmaf(__builtin_fmaf(-cz, r0, 1.f), r0, r0);     // Newton step (NaN / inf for cz = 0: geom_ok is false then)
        const float qx = cx * r, qy = cy * r;
        h.rz = r;
        h.sx = __builtin_fmaf(__builtin_fmaf(-qx, cz, cx), r, qx);                // quotient + residual / divisor: cx / cz rounded once
        h.sy = __builtin_fmaf(__builtin_fmaf(-qy, cz, cy), r, qy);
    }
    const float rho3d = h.sx * h.sx + h.sy * h.sy;
    h.dx = en.tu.w - pxf; h.dy = en.tv.w - pyf;
    const float rho2d = 2.f * (h.dx * h.dx + h.dy * h.dy);
    h.use3d = rho3d <= rho2d;
    const float rho = h.use3d ? rho3d : rho2d;
    h.depth = h.use3d ? (h.sx * en.tw.x + h.sy * en.tw.y) + en.tw.z : en.tw.z;
    {   // exp(-rho / 2) = exp2(t), t = rho * (-log2(e) / 2) carried with its rounding error and the constant's low part
        constexpr float c_hi = -0.5f * GDR_LOG2E;
        constexpr float c_lo = (float)(-0.5 * 1.4426950408889634074 - (double)c_hi);
        const float t_hi = rho * c_hi;
        const float t_lo = __builtin_fmaf(rho, c_lo, __builtin_fmaf(rho, c_hi, -t_hi));
        const float g0 = __builtin_amdgcn_exp2f(t_hi);
        h.G = __builtin_fmaf(g0, t_lo * GDR_LN2, g0);
    }
#else
    h.kx = fmaf(pxf, en.tw.x, -en.tu.x); h.ky = fmaf(pxf, en.tw.y, -en.tu.y); h.kz = fmaf(pxf, en.tw.z, -en.tu.z);
    h.lx = fmaf(pyf, en.tw.x, -en.tv.x); h.ly = fmaf(pyf, en.tw.y, -en.tv.y); h.lz = fmaf(pyf, en.tw.z, -en.tv.z);
    const float cx = h.ky * h.lz - h.kz * h.ly, cy = h.kz * h.lx - h.kx * h.lz, cz = h.kx * h.ly - h.ky * h.lx;
    h.rz = __builtin_amdgcn_rcpf(cz);
    h.sx = cx * h.rz; h.sy = cy * h.rz;
    const float rho3d = fmaf(h.sx, h.sx, h.sy * h.sy);
    h.dx = en.tu.w - pxf; h.dy = en.tv.w - pyf;
    const float rho2d = 2.f * fmaf(h.dx, h.dx, h.dy * h.dy);
    h.use3d = rho3d <= rho2d;
    const float rho = h.use3d ? rho3d : rho2d;
    h.depth = h.use3d ? fmaf(h.sx, en.tw.x, fmaf(h.sy, en.tw.y, en.tw.z)) : en.tw.z;
    h.G = __builtin_amdgcn_exp2f((-0.5f * GDR_LOG2E) * rho);
#endif
    h.alpha = fminf(0.99f, en.tw.w * h.G);
    h.geom_ok = cz != 0.f && h.depth >= GSR_NEAR;  // false for the null entry (cz = 0) and for NaNs
}

// ---------------------------------------------------------------------------------
// K6s
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(GDR_BLOCK) void surfel_render_fwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const uint32_t* __restrict__ tile_order,
    int W, int H, int gx, int ntiles, const float4* __restrict__ rec, const float* __restrict__ bg,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
    float* __restrict__ out_others, const uint32_t* __restrict__ seg_base, float* __restrict__ seg_state,
    int seg_rounds) {
    __shared__ SurfelLds lds;
    __shared__ RowLists rlists;
    __shared__ int s_done[GDR_BLOCK / GDR_WAVE];

    const uint32_t tile = tile_order ? tile_order[blockIdx.x] : xcd_remap(blockIdx.x, (uint32_t)ntiles);
    const int tx = (int)(tile % (uint32_t)gx), ty = (int)(tile / (uint32_t)gx);
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t row = lane >> 4, li = lane & 15u;
    const int sx0 = tx * GDR_TILE + (int)(wave & 1u) * 8, sy0 = ty * GDR_TILE + (int)(wave >> 1) * 8;
    const int px = sx0 + (int)(row & 1u) * 4 + (int)(li & 3u), py = sy0 + (int)(row >> 1) * 4 + (int)(li >> 2);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float XA = (float)sx0, YA = (float)sy0;
    const uint2 range = ranges[tile];
    const int total = (int)(range.y - range.x);
    const int rounds = (total + GDR_BLOCK - 1) / GDR_BLOCK;

    // cut list (render.hip, tile_order_kernel): the compositing state of every pixel is saved in front of each cut and
    // at the end of the list for K7s
    const uint32_t sb = (seg_rounds > 0 && rounds > seg_rounds) ? seg_base[tile] : 0xFFFFFFFFu;

    if (threadIdx.x == 0) null_entry(lds);
    if (threadIdx.x < 8) rlists.pad[threadIdx.x] = (uint16_t)GDR_NULL_ENTRY;
    // The distortion sum_i w_i (m_i^2 A_i + M2_i - 2 m_i M1_i) = sum_{j<i} w_i w_j (m_i - m_j)^2 only depends on
    // DIFFERENCES of the normalised depth m = far/(far-near) (1 - near/z).  Evaluated with m ~ 0.9 in fp32 it cancels
    // to ~1e-4 of its terms (the reference's loss weights it by 1000, loss.py:52); here m is taken relative to the
    // depth of the tile's first surfel, m' = K (1/z_ref - 1/z), the same constant in K6s and K7s, which keeps the
    // terms at the size of the result.  M1, M2 in final_T are sums of m'.
    const float r_ref = total > 0 ? __builtin_amdgcn_rcpf(fmaxf(rec[6 * (size_t)point_list[range.x] + 2].z, GSR_NEAR)) : 1.f;
    float thr = inside ? GDR_ALPHA_MIN : INFINITY;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f;
    float Dp = 0.f, M1 = 0.f, M2 = 0.f, dist = 0.f, med = 0.f;
    uint32_t last_contributor = 0, med_contributor = 0;

    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0, q3 = q0, q4 = q0, q5 = q0;
    bool r_valid = (int)threadIdx.x < total;
    if (r_valid) GSR_LOAD_REC(point_list[range.x + threadIdx.x]);
    for (int r = 0; r < rounds; ++r) {
        uint64_t live = __ballot(thr < INFINITY);
        if (lane == 0) s_done[wave] = live == 0ull ? 1 : 0;
        __syncthreads();
        if (s_done[0] + s_done[1] + s_done[2] + s_done[3] == GDR_BLOCK / GDR_WAVE) break;
        if (sb != 0xFFFFFFFFu && r > 0 && r % seg_rounds == 0) {  // cut in front of list position r * 256
            float* st = seg_state + ((size_t)sb + (size_t)(r / seg_rounds - 1)) * GDR_SEG_STATE_FLOATS + threadIdx.x;
            st[0] = T; st[GDR_BLOCK] = C0; st[2 * GDR_BLOCK] = C1; st[3 * GDR_BLOCK] = C2;
            st[4 * GDR_BLOCK] = N0; st[5 * GDR_BLOCK] = N1; st[6 * GDR_BLOCK] = N2;
            st[7 * GDR_BLOCK] = Dp; st[8 * GDR_BLOCK] = M1; st[9 * GDR_BLOCK] = M2;
        }
        GSR_STAGE(r_valid);
        __syncthreads();
        {
            const int nxt = (r + 1) * GDR_BLOCK + (int)threadIdx.x;
            r_valid = nxt < total;
            if (r_valid) GSR_LOAD_REC(point_list[range.x + nxt]);
        }
        if (live == 0ull) continue;
        const uint32_t base = (uint32_t)(r * GDR_BLOCK) + 1u;
        // this wave's compacted row lists of the slice (render_common.h RowLists)
        int n[4] = {0, 0, 0, 0};
        row_lists_clear(rlists, wave);
        wave_lds_fence();
#pragma unroll 1
        for (int g = 0; g < GDR_BLOCK / GDR_WAVE; ++g) {
            uint64_t m0, m1, m2, m3;
            bool mine[4];
            block_masks(lds, g, XA, YA, (live & GDR_ROW_MASK(0)) != 0ull, (live & GDR_ROW_MASK(1)) != 0ull,
                        (live & GDR_ROW_MASK(2)) != 0ull, (live & GDR_ROW_MASK(3)) != 0ull, m0, m1, m2, m3, mine);
            row_lists_append(rlists, wave, g, m0, m1, m2, m3, mine, n);
        }
        const int nmax = max(max(n[0], n[1]), max(n[2], n[3]));
        if (nmax == 0) continue;
        wave_lds_fence();
        {
            const uint16_t* my_list = &rlists.idx[wave][row][0];
            bool abort = false;
            auto fetch = [&](SEntry& en, uint32_t e) __attribute__((always_inline)) {
                en.e = e;
                en.tu = lds.tu[e]; en.tv = lds.tv[e]; en.tw = lds.tw[e]; en.nr = lds.nr[e]; en.gb = lds.gb[e];
            };
            auto composite = [&](const SEntry& en) {
                Hit h;
                intersect(en, pxf, pyf, h);
                const bool ok = h.geom_ok && h.alpha >= thr;
                const float a_c = ok ? h.alpha : 0.f;
                const float depth = ok ? h.depth : 1.f;
                const float T_new = fmaf(-a_c, T, T);
                const bool stop = T_new < 0.0001f;
                const float w = stop ? 0.f : a_c * T;
                const float A = 1.f - T;
                const float m = (GSR_FARK * GSR_NEAR) * (r_ref - __builtin_amdgcn_rcpf(depth));
                const float mm = m * m;
                dist = fmaf(fmaf(mm, A, M2) - 2.f * m * M1, w, dist);
                Dp = fmaf(depth, w, Dp);
                M1 = fmaf(m, w, M1);
                M2 = fmaf(mm, w, M2);
                const bool contributes = w > 0.f;
                const bool is_med = contributes && T > 0.5f;
                med = is_med ? depth : med;
                med_contributor = is_med ? base + en.e : med_contributor;
                N0 = fmaf(en.nr.x, w, N0); N1 = fmaf(en.nr.y, w, N1); N2 = fmaf(en.nr.z, w, N2);
                C0 = fmaf(en.nr.w, w, C0); C1 = fmaf(en.gb.x, w, C1); C2 = fmaf(en.gb.y, w, C2);
                T = stop ? T : T_new;
                thr = stop ? INFINITY : thr;
                last_contributor = contributes ? base + en.e : last_contributor;
                if (__ballot(stop) != 0ull) {
                    live = __ballot(thr < INFINITY);
                    if (live == 0ull) abort = true;
                }
            };
            SEntry A, B;   // four list positions per 8-byte LDS read; entry k+1 is fetched while entry k is composited
            uint2 q = *reinterpret_cast<const uint2*>(my_list);
            fetch(A, q.x & 0xFFFFu);
            for (int i = 0; i < nmax && !abort; i += 4) {
                const uint2 qn = *reinterpret_cast<const uint2*>(my_list + min(i + 4, GDR_BLOCK - 4));
                fetch(B, q.x >> 16);
                composite(A);
                fetch(A, q.y & 0xFFFFu);
                composite(B);
                fetch(B, q.y >> 16);
                composite(A);
                fetch(A, qn.x & 0xFFFFu);
                composite(B);
                q = qn;
            }
        }
    }
    if (sb != 0xFFFFFFFFu) {  // totals of the cut list
        const int nseg = (rounds + seg_rounds - 1) / seg_rounds;
        float* st = seg_state + ((size_t)sb + (size_t)(nseg - 1)) * GDR_SEG_STATE_FLOATS + threadIdx.x;
        st[0] = T; st[GDR_BLOCK] = C0; st[2 * GDR_BLOCK] = C1; st[3 * GDR_BLOCK] = C2;
        st[4 * GDR_BLOCK] = N0; st[5 * GDR_BLOCK] = N1; st[6 * GDR_BLOCK] = N2;
        st[7 * GDR_BLOCK] = Dp; st[8 * GDR_BLOCK] = M1; st[9 * GDR_BLOCK] = M2;
    }
    if (inside) {
        const size_t pix = (size_t)py * W + px, P = (size_t)H * W;
        final_T[pix] = T; final_T[P + pix] = M1; final_T[2 * P + pix] = M2;
        n_contrib[pix] = last_contributor; n_contrib[P + pix] = med_contributor;
        out_color[pix] = fmaf(T, bg[0], C0);
        out_color[P + pix] = fmaf(T, bg[1], C1);
        out_color[2 * P + pix] = fmaf(T, bg[2], C2);
        out_others[pix] = Dp;
        out_others[P + pix] = 1.f - T;
        out_others[2 * P + pix] = N0; out_others[3 * P + pix] = N1; out_others[4 * P + pix] = N2;
        out_others[5 * P + pix] = med;
        out_others[6 * P + pix] = dist;
    }
}

// ---------------------------------------------------------------------------------
// K7s.  grad_rec: (N,32) floats, pre-zeroed by the launcher; layout in include/gsr.h.
// ---------------------------------------------------------------------------------
// Everything K7s touches of ONE view; the kernel takes a table of V <= GDR_MAX_VIEWS of them (round 4, as render.hip's BwdViews:
// the workgroups of all views of a node in one grid, interleaved or view after view).
struct SBwdView {
    const uint2* ranges; const uint32_t* point_list; const uint32_t* tile_order;
    const float* bg; const float4* rec; const float* final_T; const uint32_t* n_contrib;
    const float* dL_dpix; const float* dL_dothers; float* grad_rec;
    const uint32_t* seg_base; const float* seg_state; const uint2* seg_extra; const uint32_t* seg_count;
    int seg_rounds, n_extra;
};
struct SBwdViews { SBwdView v[GDR_MAX_VIEWS]; };

// MAPS = false: NO view of the launch has an upstream gradient for the seven maps (dL_dothers == NULL everywhere: the
// reference's fine-stage renders and its first 1000 iterations differentiate the image only, lightning/loss.py:35-50) —
// the depth / coverage / normal / distortion recurrences and their partials drop out of the loop, dL/dz is zero.
// Record lines: a pair publishes words 0..15 with one atomic instruction and words 16..19 (normal, low-pass y) with a
// second one ONLY where a total is non-zero — the device retires ~21 G float-atomic record lines per second whatever
// they carry (scripts/ubench/atomic_probe.hip), and with two lines per pair that rate, not the VALU, bounded K7s (C5:
// 34.3 M lines per launch = 1.63 ms of 2.0; without atomics 1.57 ms).
// PAIRS: as render.hip's render_bwd_pairs_kernel — where the entries of a slice mostly cover both 4x4 blocks of a row pair, the
// two rows walk the union of their lists and publish the pair's totals once.
#define GSR_PAIR_W 6
template <bool MAPS, bool PAIRS>
__device__ __forceinline__ void surfel_render_bwd_body(const SBwdViews& vs, int V, int interleave, int n_extra_max,
                                                       int W, int H, int gx, int ntiles) {
    __shared__ SurfelLds lds;
    __shared__ RowLists rlists;
    __shared__ uint32_t s_id[GDR_BLOCK + 1];

    uint32_t view = 0, slot = blockIdx.x;      // (view, slot): as render_bwd_kernel in render.hip
    if (V > 1) {
        const uint32_t per = (uint32_t)(ntiles + n_extra_max);
        if (interleave) { view = blockIdx.x % (uint32_t)V; slot = blockIdx.x / (uint32_t)V; }
        else { view = blockIdx.x / per; slot = blockIdx.x - view * per; }
    }
    const SBwdView& bv = vs.v[view];
    const uint2* __restrict__ ranges = bv.ranges;
    const uint32_t* __restrict__ point_list = bv.point_list;
    const uint32_t* __restrict__ tile_order = bv.tile_order;
    const float* __restrict__ bg = bv.bg;
    const float4* __restrict__ rec = bv.rec;
    const float* __restrict__ final_T = bv.final_T;
    const uint32_t* __restrict__ n_contrib = bv.n_contrib;
    const float* __restrict__ dL_dpix = bv.dL_dpix;
    const float* __restrict__ dL_dothers = bv.dL_dothers;
    float* __restrict__ grad_rec = bv.grad_rec;
    const uint32_t* __restrict__ seg_base = bv.seg_base;
    const float* __restrict__ seg_state = bv.seg_state;
    const uint2* __restrict__ seg_extra = bv.seg_extra;
    const uint32_t* __restrict__ seg_count = bv.seg_count;
    const int seg_rounds = bv.seg_rounds;
    // per view: slots [0, n_extra_max): one segment of a cut list each; [n_extra_max, n_extra_max + ntiles): one tile each —
    // its whole list, or the last segment of a cut list (same scheme as render_bwd_kernel in render.hip)
    uint32_t tile;
    int seg = -1;
    if ((int)slot < n_extra_max) {
        if (slot >= min(seg_count ? seg_count[0] : 0u, (uint32_t)bv.n_extra)) return;
        const uint2 e = seg_extra[slot];
        tile = e.x;
        seg = (int)e.y;
    } else {
        const uint32_t b = slot - (uint32_t)n_extra_max;
        tile = tile_order ? tile_order[b] : xcd_remap(b, (uint32_t)ntiles);
    }
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
    // this workgroup walks list positions [seg_lo, seg_hi) of the tile, back to front
    const int full_total = (int)(range.y - range.x);
    int seg_lo = 0, seg_hi = full_total, nseg = 1;
    if (seg_rounds > 0 && full_total > seg_rounds * GDR_BLOCK && seg_base[tile] != 0xFFFFFFFFu) {
        const int seg_len = seg_rounds * GDR_BLOCK;
        nseg = (full_total + seg_len - 1) / seg_len;
        if (seg < 0) seg = nseg - 1;
        seg_lo = seg * seg_len;
        seg_hi = min(full_total, seg_lo + seg_len);
    }
    const int total = seg_hi - seg_lo;
    const uint32_t list_end = range.x + (uint32_t)seg_hi;
    const int rounds = (total + GDR_BLOCK - 1) / GDR_BLOCK;

    if (threadIdx.x == 0) { null_entry(lds); s_id[GDR_NULL_ENTRY] = 0; }
    if (threadIdx.x < 8) rlists.pad[threadIdx.x] = (uint16_t)GDR_NULL_ENTRY;
    // reference depth of the shifted normalised depth m' (see K6s): the tile's FIRST list entry
    const float r_ref = full_total > 0 ? __builtin_amdgcn_rcpf(fmaxf(rec[6 * (size_t)point_list[range.x] + 2].z, GSR_NEAR)) : 1.f;
    const float T_final = inside ? final_T[pix] : 0.f;
    const float final_D = inside ? final_T[P + pix] : 0.f, final_D2 = inside ? final_T[2 * P + pix] : 0.f;
    const float final_A = 1.f - T_final;
    float T = T_final;
    const int lc_full = inside ? (int)n_contrib[pix] : 0;
    // positions are counted from seg_lo below; a pixel whose last contributor lies behind this segment starts from
    // the state K6s saved at the cut
    const bool from_cut = lc_full > seg_hi;
    const int last_contributor = from_cut ? total : max(lc_full - seg_lo, 0);
    const int med_contributor = (inside ? (int)n_contrib[P + pix] : 0) - seg_lo;
    float gC0 = 0.f, gC1 = 0.f, gC2 = 0.f, gDepth = 0.f, gAlpha = 0.f, gN0 = 0.f, gN1 = 0.f, gN2 = 0.f, gMed = 0.f, gReg = 0.f;
    // a pixel without contributors reads no upstream gradient: the adaptor's torch ops hand NaN (0/0) to the depth
    // and alpha channels of empty pixels (renderer_2dgs.py:253-254 backward), which must not leak into a row reduction
    if (inside && last_contributor > 0) {
        gC0 = dL_dpix[pix]; gC1 = dL_dpix[P + pix]; gC2 = dL_dpix[2 * P + pix];
        if (MAPS && dL_dothers) {
            gDepth = dL_dothers[pix]; gAlpha = dL_dothers[P + pix];
            gN0 = dL_dothers[2 * P + pix]; gN1 = dL_dothers[3 * P + pix]; gN2 = dL_dothers[4 * P + pix];
            gMed = dL_dothers[5 * P + pix]; gReg = dL_dothers[6 * P + pix];
        }
    }
    const float bgT = -T_final * ((bg[0] * gC0 + bg[1] * gC1) + bg[2] * gC2);
    float B0 = 0.f, B1 = 0.f, B2 = 0.f, BD = 0.f, BA = 0.f, BN0 = 0.f, BN1 = 0.f, BN2 = 0.f, BW = 0.f;
    if (from_cut) {  // T in front of position seg_hi; every "behind" state = (total - prefix at the cut) / T
        const float* cu = seg_state + ((size_t)seg_base[tile] + (size_t)seg) * GDR_SEG_STATE_FLOATS + threadIdx.x;
        const float* to = seg_state + ((size_t)seg_base[tile] + (size_t)(nseg - 1)) * GDR_SEG_STATE_FLOATS + threadIdx.x;
        T = cu[0];
        const float rT = 1.f / T;
        B0 = (to[GDR_BLOCK] - cu[GDR_BLOCK]) * rT;
        B1 = (to[2 * GDR_BLOCK] - cu[2 * GDR_BLOCK]) * rT;
        B2 = (to[3 * GDR_BLOCK] - cu[3 * GDR_BLOCK]) * rT;
        if (MAPS) {
            BN0 = (to[4 * GDR_BLOCK] - cu[4 * GDR_BLOCK]) * rT;
            BN1 = (to[5 * GDR_BLOCK] - cu[5 * GDR_BLOCK]) * rT;
            BN2 = (to[6 * GDR_BLOCK] - cu[6 * GDR_BLOCK]) * rT;
            BD = (to[7 * GDR_BLOCK] - cu[7 * GDR_BLOCK]) * rT;
            const float dWs = T - T_final;                                  // alpha weight behind the cut
            BA = dWs * rT;
            // distortion weights dLw_j = (m_j^2 A + D2 - 2 m_j D) gReg summed with w_j over the entries behind the cut
            const float dM1 = to[8 * GDR_BLOCK] - cu[8 * GDR_BLOCK], dM2 = to[9 * GDR_BLOCK] - cu[9 * GDR_BLOCK];
            BW = (fmaf(final_A, dM2, final_D2 * dWs) - 2.f * final_D * dM1) * gReg * rT;
        }
    }

    int row_last = last_contributor;
    row_last = max(row_last, __shfl_xor(row_last, 1, 64));
    row_last = max(row_last, __shfl_xor(row_last, 2, 64));
    row_last = max(row_last, __shfl_xor(row_last, 4, 64));
    row_last = max(row_last, __shfl_xor(row_last, 8, 64));
    const int rl0 = __builtin_amdgcn_readlane(row_last, 0), rl1 = __builtin_amdgcn_readlane(row_last, 16);
    const int rl2 = __builtin_amdgcn_readlane(row_last, 32), rl3 = __builtin_amdgcn_readlane(row_last, 48);
    const int wave_last = max(max(rl0, rl1), max(rl2, rl3));

    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0, q3 = q0, q4 = q0, q5 = q0;
    uint32_t r_id = 0;
    bool r_valid = (int)threadIdx.x < total;
    if (r_valid) { r_id = point_list[list_end - 1u - threadIdx.x]; GSR_LOAD_REC(r_id); }
    for (int r = 0; r < rounds; ++r) {
        __syncthreads();
        GSR_STAGE(r_valid);
        if (r_valid) s_id[threadIdx.x] = r_id;
        __syncthreads();
        {
            const int nxt = (r + 1) * GDR_BLOCK + (int)threadIdx.x;
            r_valid = nxt < total;
            if (r_valid) { r_id = point_list[list_end - 1u - (uint32_t)nxt]; GSR_LOAD_REC(r_id); }
        }
        const int top = total - 1 - r * GDR_BLOCK;  // list position of LDS entry e: top - e
        if (top - (GDR_BLOCK - 1) >= wave_last) continue;
        int n[4] = {0, 0, 0, 0}, un[2] = {0, 0};   // this wave's compacted row lists of the slice (render_common.h RowLists)
        row_lists_clear(rlists, wave);
        wave_lds_fence();
#pragma unroll 1
        for (int g = 0; g < GDR_BLOCK / GDR_WAVE; ++g) {
            const int gtop = top - g * GDR_WAVE;  // position of this group's entry 0
            if (gtop - (GDR_WAVE - 1) >= wave_last) continue;
            uint64_t m0, m1, m2, m3;
            bool mine[4];
            const int mypos = gtop - (int)lane;
            block_masks(lds, g, XA, YA, mypos < rl0, mypos < rl1, mypos < rl2, mypos < rl3, m0, m1, m2, m3, mine);
            row_lists_append(rlists, wave, g, m0, m1, m2, m3, mine, n);
            if (PAIRS) { un[0] += __popcll(m0 | m1); un[1] += __popcll(m2 | m3); }
        }
        int nmax = max(max(n[0], n[1]), max(n[2], n[3]));
        if (nmax == 0) continue;
        bool pair_mode = false;
        if (PAIRS) {
            const int umax = max(un[0], un[1]);
            pair_mode = (umax - nmax) * GSR_PAIR_W < (n[0] + n[1] + n[2] + n[3]) - (un[0] + un[1]);
            if (pair_mode) {   // rebuild: both rows of a pair get the union of their lists
                nmax = umax;
                n[0] = n[1] = n[2] = n[3] = 0;
                wave_lds_fence();
                row_lists_clear(rlists, wave);
                wave_lds_fence();
#pragma unroll 1
                for (int g = 0; g < GDR_BLOCK / GDR_WAVE; ++g) {
                    const int gtop = top - g * GDR_WAVE;
                    if (gtop - (GDR_WAVE - 1) >= wave_last) continue;
                    uint64_t m0, m1, m2, m3;
                    bool mine[4];
                    const int mypos = gtop - (int)lane;
                    block_masks(lds, g, XA, YA, mypos < rl0, mypos < rl1, mypos < rl2, mypos < rl3, m0, m1, m2, m3, mine);
                    const bool p01 = mine[0] || mine[1], p23 = mine[2] || mine[3];
                    const bool minep[4] = {p01, p01, p23, p23};
                    row_lists_append(rlists, wave, g, m0 | m1, m0 | m1, m2 | m3, m2 | m3, minep, n);
                }
            }
        }
        // this lane's publishing unit in the hit ballot: its row, or in pair mode its row pair (the even row publishes)
        const uint64_t unit_mask = !pair_mode ? 0xFFFFull << (16 * row) : ((row & 1u) ? 0ull : 0xFFFFFFFFull << (16 * row));
        wave_lds_fence();
        {
            const uint16_t* my_list = &rlists.idx[wave][row][0];
            auto fetch = [&](SEntry& en, uint32_t e) __attribute__((always_inline)) {
                en.e = e;
                en.tu = lds.tu[e]; en.tv = lds.tv[e]; en.tw = lds.tw[e]; en.nr = lds.nr[e]; en.gb = lds.gb[e];
            };
            auto accumulate = [&](const SEntry& en) {
                Hit h;
                intersect(en, pxf, pyf, h);
                const int pos = top - (int)en.e;  // 0-based position in the tile list
                const float lim = (pos < last_contributor) ? GDR_ALPHA_MIN : INFINITY;
                const bool hit = h.geom_ok && h.alpha >= lim;
                const uint64_t hb = __ballot(hit);
                if (hb == 0ull) return;
                const float a = hit ? h.alpha : 0.f;
                const float G = hit ? h.G : 0.f, depth = hit ? h.depth : 1.f, rz = hit ? h.rz : 0.f;
                const bool u3 = hit && h.use3d;
                const float sx = u3 ? h.sx : 0.f, sy = u3 ? h.sy : 0.f;
                const float r_oma = __builtin_amdgcn_rcpf(1.f - a);
                T = T * r_oma;
                const float w = a * T;
                const float d0 = en.nr.w - B0, d1 = en.gb.x - B1, d2 = en.gb.y - B2;
                float dL_dalpha = fmaf(d0, gC0, fmaf(d1, gC1, d2 * gC2)), dL_dz = 0.f;
                B0 = fmaf(a, d0, B0); B1 = fmaf(a, d1, B1); B2 = fmaf(a, d2, B2);
                if (MAPS) {
                    const float rd = __builtin_amdgcn_rcpf(depth);
                    const float m_d = (GSR_FARK * GSR_NEAR) * (r_ref - rd);
                    const float dmd_dd = (GSR_FARK * GSR_NEAR) * rd * rd;
                    const float dLw = (fmaf(m_d * m_d, final_A, final_D2) - 2.f * m_d * final_D) * gReg;
                    const float dD = depth - BD, dA = 1.f - BA;
                    const float dn0 = en.nr.x - BN0, dn1 = en.nr.y - BN1, dn2 = en.nr.z - BN2;
                    const float dW = dLw - BW;
                    dL_dalpha = fmaf(d0, gC0, fmaf(d1, gC1, fmaf(d2, gC2, fmaf(dD, gDepth, dA * gAlpha))));
                    dL_dalpha += fmaf(dn0, gN0, fmaf(dn1, gN1, fmaf(dn2, gN2, dW)));
                    BD = fmaf(a, dD, BD); BA = fmaf(a, dA, BA);
                    BN0 = fmaf(a, dn0, BN0); BN1 = fmaf(a, dn1, BN1); BN2 = fmaf(a, dn2, BN2);
                    BW = fmaf(a, dW, BW);
                    dL_dz = fmaf(2.f * w * (m_d * final_A - final_D) * gReg, dmd_dd, w * gDepth);
                    dL_dz += (hit && pos + 1 == med_contributor) ? gMed : 0.f;
                }
                dL_dalpha = fmaf(dL_dalpha, T, bgT * r_oma);   // (only used times G below, and G = 0 without a hit)
                const float dL_dG = en.tw.w * dL_dalpha;
                const float mG = -G * dL_dG;
                // object-space branch (zero when the screen-space low-pass was taken)
                const float dsx = u3 ? fmaf(mG, sx, dL_dz * en.tw.x) : 0.f;
                const float dsy = u3 ? fmaf(mG, sy, dL_dz * en.tw.y) : 0.f;
                const float px_ = dsx * rz, py_ = dsy * rz, pz_ = -(px_ * sx + py_ * sy);
                const float dkx = h.ly * pz_ - h.lz * py_, dky = h.lz * px_ - h.lx * pz_, dkz = h.lx * py_ - h.ly * px_;
                const float dlx = py_ * h.kz - pz_ * h.ky, dly = pz_ * h.kx - px_ * h.kz, dlz = px_ * h.ky - py_ * h.kx;
                const bool lp = hit && !h.use3d;
                const float lowx = lp ? mG * 2.f * h.dx : 0.f, lowy = lp ? mG * 2.f * h.dy : 0.f;
                // record words (include/gsr.h): 0..8 dL/dT, 9 opacity, 10..12 colour, 13..14 |dTu.z|, |dTv.z|, 15 low-pass x
                const float vals[16] = {-dkx, -dky, -dkz, -dlx, -dly, -dlz,
                                        fmaf(pxf, dkx, fmaf(pyf, dlx, dL_dz * sx)), fmaf(pxf, dky, fmaf(pyf, dly, dL_dz * sy)),
                                        fmaf(pxf, dkz, fmaf(pyf, dlz, dL_dz)),
                                        G * dL_dalpha, w * gC0, w * gC1, w * gC2, fabsf(dkz), fabsf(dlz), lowx};
                float tot = row_reduce_scatter16(vals, li);
                // second line, words 16..19: normal (3), low-pass y — only the lanes whose total is not zero
                float tot4;
                if (MAPS) tot4 = row_reduce_scatter4(w * gN0, w * gN1, w * gN2, lowy, li);
                else tot4 = row_sum(lowy);
                if (PAIRS && pair_mode) { tot = rows2_sum(tot); tot4 = rows2_sum(tot4); }
                if (PAIRS ? (hb & unit_mask) != 0ull : ((hb >> (16 * row)) & 0xFFFFull) != 0ull) {
                    float* g = grad_rec + GSR_GRAD_FLOATS * (size_t)s_id[en.e];
                    atomicAdd(g + li, tot);
                    if (MAPS) { if ((li & 3u) == 0u && tot4 != 0.f) atomicAdd(g + 16 + (li >> 2), tot4); }
                    else if (li == 0u && tot4 != 0.f) atomicAdd(g + 19, tot4);
                }
            };
            SEntry A, B;
            uint2 q = *reinterpret_cast<const uint2*>(my_list);
            fetch(A, q.x & 0xFFFFu);
            for (int i = 0; i < nmax; i += 4) {
                const uint2 qn = *reinterpret_cast<const uint2*>(my_list + min(i + 4, GDR_BLOCK - 4));
                fetch(B, q.x >> 16);
                accumulate(A);
                fetch(A, q.y & 0xFFFFu);
                accumulate(B);
                fetch(B, q.y >> 16);
                accumulate(A);
                fetch(A, qn.x & 0xFFFFu);
                accumulate(B);
                q = qn;
            }
        }
    }
}

template <bool MAPS>
__global__ __launch_bounds__(GDR_BLOCK) void surfel_render_bwd_kernel(const SBwdViews vs, int V, int interleave, int n_extra_max,
                                                                      int W, int H, int gx, int ntiles) {
    surfel_render_bwd_body<MAPS, false>(vs, V, interleave, n_extra_max, W, H, gx, ntiles);
}
template <bool MAPS>
__global__ __launch_bounds__(GDR_BLOCK) void surfel_render_bwd_pairs_kernel(const SBwdViews vs, int V, int interleave,
                                                                            int n_extra_max, int W, int H, int gx, int ntiles) {
    surfel_render_bwd_body<MAPS, true>(vs, V, interleave, n_extra_max, W, H, gx, ntiles);
}

}  // namespace

static thread_local int t_sbwd_pairs = 0;
void surfel_render_bwd_set_pairs(int pairs) { t_sbwd_pairs = pairs; }

hipError_t launch_surfel_render_fwd(const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                    const gdr_image* img, const gsr_outputs* out, hipStream_t st) {
    const int W = s->image_width, H = s->image_height;
    const int gx = tile_grid_x(W), gy = tile_grid_y(H);
    const int ntiles = gx * gy;
    GDR_LAUNCH(GDR_K_RENDER_FWD, surfel_render_fwd_kernel, dim3(ntiles), dim3(GDR_BLOCK), st, (const uint2*)img->ranges,
               bin->values[bin->sorted], img->tile_order, W, H, gx, ntiles, (const float4*)g->rec, s->bg, img->final_T,
               img->n_contrib, out->color, out->allmap, GDR_SEG_FWD_ARGS(bin, img));
    return hipGetLastError();
}

// The (N,4) means2D gradient of ONE view from its K7s gradient record (what K9s forms inside its per-view loop:
// preprocess_surfel.hip): the densification signal dL/dTu.z, dL/dTv.z x depth x W/2 | H/2 in columns 0-1 and its
// per-pixel-|.| twin (record words 13, 14) in columns 2-3; zero for culled surfels.  Used by the render groups
// (viewgroup.py): every call of the unchanged caller owns a carrier and gets ITS view's gradient, while the group's one
// K9s returns the sums over the views for the shared inputs.
namespace {
__global__ __launch_bounds__(GDR_BLOCK) void surfel_means2d_view_kernel(int N, const float4* __restrict__ grad_rec,
                                                                        const float4* __restrict__ rec,
                                                                        const int32_t* __restrict__ radii, float hw, float hh,
                                                                        float4* __restrict__ out) {
    const int i = blockIdx.x * GDR_BLOCK + threadIdx.x;
    if (i >= N) return;
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
    if (radii[i] > 0) {
        const float4 g0 = grad_rec[8 * (size_t)i], g1 = grad_rec[8 * (size_t)i + 1], g3 = grad_rec[8 * (size_t)i + 3];
        const float depth = rec[6 * (size_t)i + 2].z;
        m = make_float4(g0.z * depth * hw, g1.y * depth * hh, g3.y * depth * hw, g3.z * depth * hh);
    }
    out[i] = m;
}
}  // namespace

hipError_t launch_surfel_means2d_view(int N, const gdr_settings* s, const gdr_geom* g, const int32_t* radii,
                                      const float* grad_rec, float* out, hipStream_t st) {
    GDR_LAUNCH(GDR_K_SURFEL_MAPS, surfel_means2d_view_kernel, dim3((N + GDR_BLOCK - 1) / GDR_BLOCK), dim3(GDR_BLOCK), st, N,
               (const float4*)grad_rec, (const float4*)g->rec, radii, 0.5f * (float)s->image_width,
               0.5f * (float)s->image_height, (float4*)out);
    return hipGetLastError();
}

// K7s of V <= GDR_MAX_VIEWS views of one image size in ONE launch (gsr_render_backward_views); V = 1: the single-view entry
hipError_t launch_surfel_render_bwd_views(int V, const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                          const gdr_image* img, const gsr_grad_inputs* gi, float* const* grad_recs, int interleave,
                                          hipStream_t st) {
    const int W = s[0].image_width, H = s[0].image_height;
    const int gx = tile_grid_x(W), gy = tile_grid_y(H);
    const int ntiles = gx * gy;
    SBwdViews vs;
    memset(&vs, 0, sizeof(vs));
    int n_extra_max = 0;
    for (int v = 0; v < V; ++v) {
        SBwdView& b = vs.v[v];
        b.ranges = (const uint2*)img[v].ranges; b.point_list = bin[v].values[bin[v].sorted]; b.tile_order = img[v].tile_order;
        b.bg = s[v].bg; b.rec = (const float4*)g[v].rec; b.final_T = img[v].final_T; b.n_contrib = img[v].n_contrib;
        b.dL_dpix = gi[v].dL_dcolor; b.dL_dothers = gi[v].dL_dallmap; b.grad_rec = grad_recs[v];
        b.seg_rounds = seg_rounds_of(&bin[v], &img[v]);
        b.n_extra = b.seg_rounds ? bin[v].seg_cap : 0;
        b.seg_base = img[v].seg_base; b.seg_state = (const float*)bin[v].seg_state;
        b.seg_extra = (const uint2*)bin[v].seg_extra; b.seg_count = bin[v].seg_count;
        n_extra_max = b.n_extra > n_extra_max ? b.n_extra : n_extra_max;
    }
    bool maps = false;
    for (int v = 0; v < V; ++v) maps = maps || gi[v].dL_dallmap != nullptr;
    const dim3 grid((unsigned)V * (unsigned)(ntiles + n_extra_max));
    if (maps && t_sbwd_pairs)
        GDR_LAUNCH(GDR_K_RENDER_BWD, surfel_render_bwd_pairs_kernel<true>, grid, dim3(GDR_BLOCK), st, vs, V, interleave, n_extra_max, W, H, gx, ntiles);
    else if (maps)
        GDR_LAUNCH(GDR_K_RENDER_BWD, surfel_render_bwd_kernel<true>, grid, dim3(GDR_BLOCK), st, vs, V, interleave, n_extra_max, W, H, gx, ntiles);
    else if (t_sbwd_pairs)
        GDR_LAUNCH(GDR_K_RENDER_BWD, surfel_render_bwd_pairs_kernel<false>, grid, dim3(GDR_BLOCK), st, vs, V, interleave, n_extra_max, W, H, gx, ntiles);
    else
        GDR_LAUNCH(GDR_K_RENDER_BWD, surfel_render_bwd_kernel<false>, grid, dim3(GDR_BLOCK), st, vs, V, interleave, n_extra_max, W, H, gx, ntiles);
    return hipGetLastError();
}

hipError_t launch_surfel_render_bwd(const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                    const gdr_image* img, const gsr_grad_inputs* gi, float* grad_rec,
                                    hipStream_t st) {
    return launch_surfel_render_bwd_views(1, s, g, bin, img, gi, &grad_rec, 0, st);
}

}  // namespace gdr
