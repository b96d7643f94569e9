// render.hip — K6 (per-tile alpha-composited forward) and K7 (per-pixel reverse-order
// backward) of the rasterizer for gfx950, plus the tile-order helper.
// Behaviour: SURVEY.md Appendix A.3 / A.4 (3DGS tile renderer + depth / alpha outputs
// + AbsGS |.|-accumulated screen-space gradients), i.e. what the reference obtains from
// rasterizer(...) at /root/reference/lightning/renderer.py:250-259 and differentiates at
// /root/reference/lightning/network.py:867-878.
//
// CDNA4 mapping (not the 32-wide warp layout of the CUDA lineage), each point measured
// (profiles/):
//   * one workgroup = one 16x16 tile = 4 wavefronts; a wave owns an 8x8 sub-tile and each of
//     its four 16-lane DPP rows composites its OWN 4x4 pixel block against its OWN culled
//     sub-list, so one wave instruction advances up to four different Gaussians;
//   * every Gaussian's render attributes live in ONE 64-byte record (xy, conic + opacity,
//     rgb + depth, alpha-extent; written by K1), so the gather by sorted id touches one cache
//     line per entry instead of three; the tile's sorted slice is staged 256 entries at a time
//     in LDS while the next slice is already being fetched into registers; every lane tests ONE staged entry's {alpha >= 1/255} bounding box
//     against the four blocks and four 64-bit ballots give each block its sub-list (SGPR masks);
//     the test is conservative, so results (incl. n_contrib = position in the FULL tile list)
//     are identical to evaluating every entry;
//   * the scalar unit is shared by the CU's four SIMDs, so the inner loop keeps SALU work to
//     the mask walk (s_ff1 / s_bitset0 per row): predication is done with FLOAT thresholds
//     (a finished pixel's alpha threshold becomes +inf; idle rows read a null LDS entry with
//     opacity 0) instead of SGPR mask algebra; LDS reads for the next entry are issued before
//     the current one is composited;
//   * G = v_exp_f32 on a conic pre-scaled by log2(e); forward and backward stage identically,
//     so both make identical skip decisions;
//   * backward: the "colour behind" recurrences are kept in the form B <- B + a (c - B), which
//     leaves the state untouched when a = 0 (no selects); the 12 per-Gaussian partial gradients
//     are reduce-scattered inside each 16-lane row with DPP (one total per lane) and published
//     with ONE global_atomic_add_f32 instruction into a 64-byte per-Gaussian record; |.| of the
//     mean2D terms is taken per pixel BEFORE the reduction (AbsGS semantics);
//   * workgroups take tiles longest-list-first (tile_order), which removes the tail of a few
//     heavy tiles; without an order the blockIdx -> tile mapping is XCD-aware.
#include <stdlib.h>
#include <string.h>

#include "gdr_common.h"
#include "render_common.h"

namespace gdr {

namespace {

// exponent (in log2 units: the conic carries the log2(e) factor) shared by forward and
// backward: explicit operation order and explicit fused multiply-adds.
__device__ __forceinline__ float gauss_power(float dx, float dy, float cx, float cy, float cz) {
#pragma clang fp contract(off)
    const float s = fmaf(cz, dy * dy, cx * (dx * dx));
    return fmaf(-0.5f, s, -(cy * (dx * dy)));
}

// ---- threshold guard (round 5; measured, NOT in the product build: GDR_THRESHOLD_GUARD = 0) ------------------------------
// The fast path evaluates alpha = opacity * v_exp_f32(p2) on a conic pre-scaled by log2(e) with fused multiply-adds; the
// oracle (oracle/gdr_oracle.c:277-283) evaluates min(0.99, opacity * expf(power)), power = -0.5 (a dx dx + c dy dy) - b dx dy,
// with glibc's expf.  Two fp32 evaluations of one formula: they differ by a few ulp — irrelevant to the images (4e-7) except
// where alpha lands within that distance of the 1/255 skip threshold: there they decide differently and a pixel gains or
// loses a contributor of weight ~1/255 (the 1-5 "threshold pixels" per four 800x800 views: max |dRGB| 1.6e-3 on them).
// The guard sends a lane whose fast alpha is within GDR_ALPHA_BAND of 1/255 (32 ulp) through a slow path — the Gaussian's
// record re-read, the oracle's operation order without contraction, the full-precision expf — shared by K6, the deep K6 and
// K7.  Same-box A/B (profiles/r05_ab_threshold_guard.txt): it settles 2 of the 5 flipped pixels of the four C4 views and
// costs C4 -2.0 %, C2 -4.0 %, C3 -3.2 % (K6 +5..12 %, K7 +2..4 %: the branch behind every evaluation breaks the
// fetch / composite interleave of the unrolled walk).  The other 3 are out of its reach: for elongated Gaussians the terms of
// p2 are ~200 and cancel to ~8, so the two evaluations differ by up to ~2e-5 relative — a band that wide is no longer rare
// (one tile slice in four) —, and the `T < 1e-4` stop carries the accumulated rounding of all earlier factors.  The CUDA
// reference's own expf differs from glibc's in the same way; a flip is the difference between two valid fp32 programs, not
// an error of either.  Compile with -DGDR_THRESHOLD_GUARD=1 to study it.
#define GDR_ALPHA_BAND 1.5e-8f
#ifndef GDR_THRESHOLD_GUARD
#define GDR_THRESHOLD_GUARD 0
#endif
#if GDR_THRESHOLD_GUARD
#define GDR_WALK_WAVES __attribute__((amdgpu_waves_per_eu(5, 5)))   /* the guard's rare path may spill; the walk keeps 5 waves / SIMD */
__device__ __forceinline__ bool alpha_near_threshold(float alpha) { return fabsf(alpha - GDR_ALPHA_MIN) <= GDR_ALPHA_BAND; }
#else
#define GDR_WALK_WAVES
__device__ __forceinline__ bool alpha_near_threshold(float) { return false; }
#endif
// occupancy experiments of the measurement builds (round 6): -DGDR_K6_WAVES=n / -DGDR_K7_WAVES=n pin the waves per SIMD of the
// standard K6 / the row-mode K7 (the compiler then caps their VGPRs at 512 / n); not set in the product build
#ifdef GDR_K6_WAVES
#define GDR_K6_OCC __attribute__((amdgpu_waves_per_eu(GDR_K6_WAVES, GDR_K6_WAVES)))
#else
#define GDR_K6_OCC
#endif
#ifdef GDR_K7_WAVES
#define GDR_K7_OCC __attribute__((amdgpu_waves_per_eu(GDR_K7_WAVES, GDR_K7_WAVES)))
#else
#define GDR_K7_OCC
#endif
// ---- measurement builds (round 6: the K7 instruction / time budget, profiles/r06_k7_budget.json) --------------------------
// -DGDR_K7_STUB=<bits> compiles ONE phase of K7's walk out (or twice in) so that its share of the launch can be MEASURED as
// a difference of launch times and SQ_INSTS_VALU counts (scripts/gpu_k7_budget.sh).  Results of such a build are WRONG by
// construction: the Makefile gives it a non-"release" build tag, which the Python loader refuses by default.
//   1  the record atomics are never executed (the publish branch is kept, its condition is never true)
//   2  the 12-value DPP reduce-scatter is replaced by a plain 11-add chain (every value stays live)
//   4  the gradient terms behind dL/dalpha are skipped (all 12 values = dL/dalpha)
//   8  the slice cull (block_masks + row_lists_append) runs TWICE (cost of the cull = this build - the product build)
//  16  the "behind" recurrences and the dL/dalpha dot product are skipped (dL/dalpha = T)
//  32  K6: the slice cull runs twice
#ifndef GDR_K7_STUB
#define GDR_K7_STUB 0
#endif
__device__ __forceinline__ float2 alpha_exact(const float4* __restrict__ rec, uint32_t id, float pxf, float pyf) {
#pragma clang fp contract(off)
    const float4 a0 = rec[4 * (size_t)id], co = rec[4 * (size_t)id + 1];
    const float dx = a0.x - pxf, dy = a0.y - pyf;
    const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
    const float G = expf(power);
    const float alpha = fminf(0.99f, co.w * G);
    return make_float2(power > 0.f ? 0.f : alpha, G);     // (alpha, G)
}


struct Entry {  // one staged list entry, in registers
    uint32_t e;
    float2 m;
    float4 co, cd;
};

// LDS image of one 256-entry slice (+ the null entry)
struct SliceLds {
    float2 xy[GDR_BLOCK + 1];
    float2 ext[GDR_BLOCK];
    float4 co[GDR_BLOCK + 1];
    float4 cd[GDR_BLOCK + 1];
};

__device__ __forceinline__ void stage_write(SliceLds& s, bool valid, float4 xe, float4 co, float4 cd) {
    if (valid) {  // xe = (x, y, extent_x, extent_y)
        s.xy[threadIdx.x] = make_float2(xe.x, xe.y);
        s.ext[threadIdx.x] = make_float2(xe.z, xe.w);
        s.co[threadIdx.x] = make_float4(co.x * GDR_LOG2E, co.y * GDR_LOG2E, co.z * GDR_LOG2E, co.w);
        s.cd[threadIdx.x] = cd;
    } else {
        s.ext[threadIdx.x] = make_float2(-1.f, -1.f);  // culled for every block
        s.xy[threadIdx.x] = make_float2(0.f, 0.f);
    }
}

// per-block overlap ballots of the 64 staged entries [g*64, g*64+64) against the wave's four
// 4x4 blocks; `allow` is an extra per-lane condition (backward: list position)
__device__ __forceinline__ void block_masks(const SliceLds& s, int g, float XA, float YA, bool a0, bool a1,
                                            bool a2, bool a3, uint64_t& m0, uint64_t& m1, uint64_t& m2,
                                            uint64_t& m3, bool (&mine)[4]) {
    const float2 m = s.xy[g * GDR_WAVE + (int)lane_id()];
    const float2 h = s.ext[g * GDR_WAVE + (int)lane_id()];
    const bool v = h.x >= 0.f;
    const float lo_x = m.x - h.x, hi_x = m.x + h.x, lo_y = m.y - h.y, hi_y = m.y + h.y;
    const bool x0 = v && hi_x >= XA && lo_x <= XA + 3.f, x1 = v && hi_x >= XA + 4.f && lo_x <= XA + 7.f;
    const bool y0 = hi_y >= YA && lo_y <= YA + 3.f, y1 = hi_y >= YA + 4.f && lo_y <= YA + 7.f;
    mine[0] = x0 && y0 && a0; mine[1] = x1 && y0 && a1; mine[2] = x0 && y1 && a2; mine[3] = x1 && y1 && a3;
    m0 = __ballot(mine[0]);
    m1 = __ballot(mine[1]);
    m2 = __ballot(mine[2]);
    m3 = __ballot(mine[3]);
}

// ---------------------------------------------------------------------------------
// tile order: tiles sorted by descending list length (counting sort on length/16, one
// workgroup).  order[k] = tile id of the k-th workgroup.
// ---------------------------------------------------------------------------------
#define GDR_ORDER_BUCKETS 1024
// Cut-list tables (seg_base != nullptr, seg_len > 0): a tile whose list is longer than seg_len entries is cut every
// seg_len entries into nseg segments.  seg_base[tile] = its first slot in seg_state (nseg slots),
// 0xFFFFFFFF for uncut tiles; seg_extra = (tile, segment) of every segment but the last of its tile (K7: the last one
// is walked by the tile's own workgroup); seg_count = {rows of seg_extra, slots}.  Sum nseg <= 2 D / seg_len: the
// tables (capacity seg_cap rows, 2 * seg_cap slots) cannot overflow; the guards are belt and braces.
#define GDR_ORDER_THREADS 1024   // one workgroup per view, sixteen waves: every phase below is a few trips over the tiles
__global__ __launch_bounds__(GDR_ORDER_THREADS) void tile_order_kernel(const BinViews vs, int ntiles) {
    constexpr int TB = GDR_ORDER_THREADS, NWV = TB / GDR_WAVE;
    const BinView& bv = vs.v[blockIdx.y];   // one workgroup per view
    const uint2* ranges = bv.ranges;        // (not __restrict__: with from_totals this kernel writes them first)
    uint32_t* __restrict__ order = bv.tile_order;
    const int seg_len = bv.seg_len, seg_cap = bv.seg_cap;
    uint32_t* __restrict__ seg_base = bv.seg_base;
    uint2* __restrict__ seg_extra = bv.seg_extra;
    uint32_t* __restrict__ seg_count = bv.seg_count;
    const uint32_t deep_max_busy = bv.deep_max_busy;
    __shared__ uint32_t cnt[GDR_ORDER_BUCKETS];
    __shared__ uint32_t wsum[NWV];
    if (bv.from_totals) {   // direct tile binning: the per-tile totals (behind the count matrix) -> ranges, empty tiles (0,0);
        // clamped to the capacity of a device-sized call
        const uint32_t* __restrict__ totals = bv.tile_hist + (size_t)bv.hist_width * (size_t)((ntiles + 63) / 64 * 64);
        const uint32_t cap = (uint32_t)(bv.D < 0xFFFFFFFFull ? bv.D : 0xFFFFFFFFull);
        uint2* __restrict__ rw = bv.ranges;
        uint32_t carry = 0u;
        for (int t0 = 0; t0 < ntiles; t0 += TB) {
            const int t = t0 + (int)threadIdx.x;
            const uint32_t c = t < ntiles ? totals[t] : 0u;
            uint32_t incl = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t u = __shfl_up(incl, off, 64);
                if ((int)lane_id() >= off) incl += u;
            }
            __syncthreads();
            if (lane_id() == 63) wsum[threadIdx.x >> 6] = incl;
            __syncthreads();
            uint32_t base = carry + incl - c, tot = 0u;
            for (uint32_t w = 0; w < (uint32_t)NWV; ++w) {
                if (w < (threadIdx.x >> 6)) base += wsum[w];
                tot += wsum[w];
            }
            if (t < ntiles) rw[t] = c ? make_uint2(min(base, cap), min(base + c, cap)) : make_uint2(0u, 0u);
            carry += tot;
        }
        __threadfence_block();
        __syncthreads();
    }
    for (int k = threadIdx.x; k < GDR_ORDER_BUCKETS; k += TB) cnt[k] = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < ntiles; t += TB) {
        const uint2 r = ranges[t];
        const uint32_t b = min((r.y - r.x) >> 4, (uint32_t)GDR_ORDER_BUCKETS - 1u);
        atomicAdd(&cnt[GDR_ORDER_BUCKETS - 1 - b], 1u);  // bucket 0 = longest lists
    }
    __syncthreads();
    // exclusive scan of the 1024 buckets: one per thread
    static_assert(GDR_ORDER_BUCKETS == TB, "one bucket per thread");
    {
        const uint32_t s = cnt[threadIdx.x];
        uint32_t incl = s;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if ((int)lane_id() >= off) incl += t;
        }
        if (lane_id() == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t base = incl - s;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) base += wsum[w];
        cnt[threadIdx.x] = base;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ntiles; t += TB) {
        const uint2 r = ranges[t];
        const uint32_t b = min((r.y - r.x) >> 4, (uint32_t)GDR_ORDER_BUCKETS - 1u);
        order[atomicAdd(&cnt[GDR_ORDER_BUCKETS - 1 - b], 1u)] = (uint32_t)t;
    }
    // what the caller may feed back into the next call of this scene shape (gdr_binning.stats_out): tiles in the tile
    // sort's long / medium class, busy tiles (lists of >= 64 entries)
    uint32_t n_long = 0u, n_medium = 0u, n_busy = 0u, busy_len = 0u;   // busy_len: entries in the busy tiles' lists, / 16
    for (int t = threadIdx.x; t < ntiles; t += TB) {
        const uint2 r = ranges[t];
        const uint32_t len = r.y - r.x;
        n_long += len > (uint32_t)GDR_TSORT_MEDIUM ? 1u : 0u;
        n_medium += (len > (uint32_t)GDR_TSORT_SMALL && len <= (uint32_t)GDR_TSORT_MEDIUM) ? 1u : 0u;
        n_busy += len >= 64u ? 1u : 0u;
        busy_len += len >= 64u ? len >> 4 : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n_long += __shfl_xor(n_long, off, 64); n_medium += __shfl_xor(n_medium, off, 64); n_busy += __shfl_xor(n_busy, off, 64);
        busy_len += __shfl_xor(busy_len, off, 64);
    }
    __syncthreads();   // (cnt is free again: the ordering phase is over)
    if (lane_id() == 0) {
        cnt[threadIdx.x >> 6] = n_long; cnt[NWV + (threadIdx.x >> 6)] = n_medium; cnt[2 * NWV + (threadIdx.x >> 6)] = n_busy;
        cnt[3 * NWV + (threadIdx.x >> 6)] = busy_len;
    }
    __syncthreads();
    n_long = n_medium = n_busy = busy_len = 0u;
    for (int w = 0; w < NWV; ++w) { n_long += cnt[w]; n_medium += cnt[NWV + w]; n_busy += cnt[2 * NWV + w]; busy_len += cnt[3 * NWV + w]; }
    // "deep" forward: few busy tiles AND long lists in them (mean >= GDR_DEEP_MIN_MEAN entries).  Few busy tiles with
    // short lists (C3: 700 of 1024 tiles with ~1.4 k entries, C2 `shell`: 530 tiles with ~1.3 k) are a handful of rounds per
    // workgroup, and the standard kernel's half as many instructions win: C3 2995 -> 3110, C2 `shell` 2938 -> 3060 views/s
    // without the deep launch; with lists of 4 k (C3 `shell`) and 7.6 k (C4 `shell`) it is worth +14 % and +3 %.
    const uint32_t deep = (seg_base != nullptr && seg_len > 0 && n_busy <= deep_max_busy &&
                           (uint64_t)busy_len * 16ull >= (uint64_t)n_busy * (uint64_t)bv.deep_min_mean) ? 1u : 0u;
    if (threadIdx.x == 0 && bv.stats_out) {
        bv.stats_out[0] = n_long; bv.stats_out[1] = n_medium; bv.stats_out[2] = deep; bv.stats_out[3] = n_busy;
    }
    __syncthreads();
    if (seg_base == nullptr) return;
    if (seg_len <= 0) {
        if (threadIdx.x < 3) seg_count[threadIdx.x] = 0u;
        return;
    }
    {   // seg_count[2] = "deep" flag: few busy tiles (an object in front of an empty background) -> the forward of the
        // cut tiles runs with 16 instead of 64 pixels per wave (render_fwd_deep_kernel).  Busy = list of >= 64 entries;
        // the chip holds 1024 forward workgroups at once (256 CUs x 4), so below ~768 busy tiles CUs sit on one
        // workgroup (one wave per SIMD) and the walk of a long list is latency-bound.
        if (threadIdx.x == 0) seg_count[2] = deep;
    }
    uint32_t slot_run = 0u, cut_run = 0u;  // uniform running totals over the chunks of TB tiles
    for (int t0 = 0; t0 < ntiles; t0 += TB) {
        const int t = t0 + (int)threadIdx.x;
        uint32_t nseg = 0u;
        if (t < ntiles) {
            const uint2 r = ranges[t];
            const uint32_t len = r.y - r.x;
            if (len > (uint32_t)seg_len) nseg = (len + (uint32_t)seg_len - 1u) / (uint32_t)seg_len;
        }
        const uint32_t cut = nseg ? 1u : 0u;
        uint32_t inc = nseg, cinc = cut;  // block-wide inclusive scans of (slots, cut tiles)
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t u = __shfl_up(inc, off, 64), v = __shfl_up(cinc, off, 64);
            if ((int)lane_id() >= off) { inc += u; cinc += v; }
        }
        __syncthreads();  // the previous chunk's / the ordering phase's reads of wsum, cnt are done
        if (lane_id() == 63) { wsum[threadIdx.x >> 6] = inc; cnt[threadIdx.x >> 6] = cinc; }
        __syncthreads();
        uint32_t sbase = inc - nseg, cbase = cinc - cut, stot = 0u, ctot = 0u;
        for (uint32_t w = 0; w < (uint32_t)NWV; ++w) {
            if (w < (threadIdx.x >> 6)) { sbase += wsum[w]; cbase += cnt[w]; }
            stot += wsum[w]; ctot += cnt[w];
        }
        if (t < ntiles) {
            const uint32_t slot = slot_run + sbase;             // slots before this tile
            const uint32_t e0 = slot - (cut_run + cbase);       // rows of seg_extra before it: one less per cut tile
            const bool fits = nseg && slot + nseg <= 2u * (uint32_t)seg_cap && e0 + nseg - 1u <= (uint32_t)seg_cap;
            seg_base[t] = fits ? slot : 0xFFFFFFFFu;
            for (uint32_t k = 0; fits && k + 1u < nseg; ++k) seg_extra[e0 + k] = make_uint2((uint32_t)t, k);
        }
        slot_run += stot;
        cut_run += ctot;
    }
    if (threadIdx.x == 0) {
        seg_count[0] = min(slot_run - cut_run, (uint32_t)seg_cap);
        seg_count[1] = min(slot_run, 2u * (uint32_t)seg_cap);
    }
}

// ---------------------------------------------------------------------------------
// K6
// ---------------------------------------------------------------------------------
// LOSS: the image loss of the measurement / training step folded into the epilogue:
//   loss += mean_{c,p}(clamp(color,0,1) - target)^2 + w_depth mean(depth) + w_alpha mean(alpha)
// (renderer.py:261 clamp, loss.py:37-38 MSE; depth / alpha means: §8d) — one atomicAdd per tile.
struct FusedLoss {
    const float* target;  // (3,H,W)
    float w_depth, w_alpha;
    float* loss;          // K6: accumulated (caller zeroes)
    const float* go;      // K7: upstream scalar gradient (device)
    const float* color;   // K7: the colour K6 wrote, (3,H,W)
    float go_scale;       // K6 LOSS = 2: host-known upstream scalar (1 / views for the mean over views)
};

// Cut lists (gdr_binning.seg_len, tables built by tile_order_kernel): a tile list longer than seg_rounds slices is
// cut every seg_rounds slices; in front of every cut and at the end of the list the kernel saves each pixel's
// compositing state (T and the colour / depth / alpha sums so far) so that K7 can walk the segments of one tile in
// parallel workgroups.  The forward itself stays one workgroup per tile: compositing every segment speculatively
// from T = 1 in parallel workgroups (colour is linear in the incoming transmittance) and walking only the segments
// in which a pixel saturates was built and measured — it doubles the alpha evaluations of cut lists and lost 4-16 %
// once two views are in flight and the backward is segmented.
// LOSS = 2 (the abs-grad-only path of network.py:865-878): the MSE is folded in as for LOSS = 1 but NO image
// leaves the kernel — instead of colour / depth / alpha it writes d loss / d colour of the pixel (times go_scale) into
// out_color, which the mean2D-only K7 reads as its upstream gradient; per pixel 12 + 8 bytes instead of 20 + 8.
// Everything K6 touches of ONE view; the kernel takes a table of V <= GDR_MAX_VIEWS of them (round 4, as K7's BwdViews:
// the workgroups of all views of a node in one grid).
struct FwdView {
    const uint2* ranges; const uint32_t* point_list; const uint32_t* tile_order;
    const float4* rec; const float* bg; float* final_T; uint32_t* n_contrib;
    float* out_color; float* out_depth; float* out_alpha;
    FusedLoss fl;
    const uint32_t* seg_base; float* seg_state; const uint32_t* deep_flag;
    int seg_rounds, pad;
};
struct FwdViews { FwdView v[GDR_MAX_VIEWS]; };

template <int LOSS>
__global__ __launch_bounds__(GDR_BLOCK) GDR_WALK_WAVES GDR_K6_OCC
void render_fwd_kernel(const FwdViews vs, int V, int interleave, int W, int H, int gx,
                                                               int ntiles) {
    __shared__ SliceLds lds;
    __shared__ RowLists rlists;
    __shared__ int s_done[GDR_BLOCK / GDR_WAVE];
    uint32_t view = 0, slot = blockIdx.x;     // (view, slot): as render_bwd_kernel
    if (V > 1) {
        if (interleave) { view = blockIdx.x % (uint32_t)V; slot = blockIdx.x / (uint32_t)V; }
        else { view = blockIdx.x / (uint32_t)ntiles; slot = blockIdx.x - view * (uint32_t)ntiles; }
    }
    const FwdView& fv = vs.v[view];
    const uint2* __restrict__ ranges = fv.ranges;
    const uint32_t* __restrict__ point_list = fv.point_list;
    const uint32_t* __restrict__ tile_order = fv.tile_order;
    const float4* __restrict__ rec = fv.rec;
    const float* __restrict__ bg = fv.bg;
    float* __restrict__ final_T = fv.final_T;
    uint32_t* __restrict__ n_contrib = fv.n_contrib;
    float* __restrict__ out_color = fv.out_color;
    float* __restrict__ out_depth = fv.out_depth;
    float* __restrict__ out_alpha = fv.out_alpha;
    const FusedLoss& fl = fv.fl;
    const uint32_t* __restrict__ seg_base = fv.seg_base;
    float* __restrict__ seg_state = fv.seg_state;
    const int seg_rounds = fv.seg_rounds;
    const uint32_t* __restrict__ deep_flag = fv.deep_flag;

    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t row = lane >> 4, li = lane & 15u;
    if (threadIdx.x == 0) {
        lds.xy[GDR_NULL_ENTRY] = make_float2(0.f, 0.f);
        lds.co[GDR_NULL_ENTRY] = make_float4(0.f, 0.f, 0.f, 0.f);
        lds.cd[GDR_NULL_ENTRY] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (threadIdx.x < 8) rlists.pad[threadIdx.x] = (uint16_t)GDR_NULL_ENTRY;

    const uint32_t tile = tile_order ? tile_order[slot] : xcd_remap(slot, (uint32_t)ntiles);
    const int tx = (int)(tile % (uint32_t)gx), ty = (int)(tile / (uint32_t)gx);
    const int sx0 = tx * GDR_TILE + (int)(wave & 1u) * 8, sy0 = ty * GDR_TILE + (int)(wave >> 1) * 8;
    const int px = sx0 + (int)(row & 1u) * 4 + (int)(li & 3u), py = sy0 + (int)(row >> 1) * 4 + (int)(li >> 2);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float XA = (float)sx0, YA = (float)sy0;  // block k: x in [XA+4(k&1), +3], y in [YA+4(k>>1), +3]
    const uint2 range = ranges[tile];
    const int full_total = (int)(range.y - range.x);
    const int full_rounds = (full_total + GDR_BLOCK - 1) / GDR_BLOCK;
    const uint32_t sb = (seg_rounds > 0 && full_rounds > seg_rounds) ? seg_base[tile] : 0xFFFFFFFFu;
    const bool cut = sb != 0xFFFFFFFFu;
    if (cut && deep_flag && *deep_flag) return;  // rendered by render_fwd_deep_kernel (launched in front of this one)
    const int nseg = cut ? (full_rounds + seg_rounds - 1) / seg_rounds : 1;

    // a pixel contributes while alpha >= thr; thr = +inf once it is saturated ("done") or outside
    float thr = inside ? GDR_ALPHA_MIN : INFINITY;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, Wt = 0.f;
    uint32_t last_contributor = 0;

    // One loop over the 256-entry rounds of the whole list.  The gather of a slice is two dependent trips to memory
    // (sorted id -> 64-byte record): the ids are fetched TWO rounds ahead and the records one round ahead, so that neither
    // is waited for at the top of a round.  (Until round 3 the list was walked segment by segment, each walk with its own
    // cold gather — with seg_len = 256 that was every round.)
    auto load_rec = [&](uint32_t id, float4& xe, float4& co, float4& cd) __attribute__((always_inline)) {
        const float4 a0 = rec[4 * (size_t)id], a3 = rec[4 * (size_t)id + 3];
        xe = make_float4(a0.x, a0.y, a3.x, a3.y); co = rec[4 * (size_t)id + 1]; cd = rec[4 * (size_t)id + 2];
    };
    float4 r_xe = make_float4(0.f, 0.f, 0.f, 0.f), r_co = r_xe, r_cd = r_xe;
    bool r_valid = (int)threadIdx.x < full_total;
    if (r_valid) load_rec(point_list[range.x + threadIdx.x], r_xe, r_co, r_cd);
    bool n_valid = GDR_BLOCK + (int)threadIdx.x < full_total;
    uint32_t n_id = n_valid ? point_list[range.x + (uint32_t)GDR_BLOCK + threadIdx.x] : 0u;
    for (int r = 0; r < full_rounds; ++r) {
        const int pos0 = r * GDR_BLOCK;
        if (cut && r > 0 && r % seg_rounds == 0) {  // cut in front of list position pos0: the state K7 starts the earlier segments from
            float* st = seg_state + ((size_t)sb + (size_t)(r / seg_rounds - 1)) * GDR_SEG_STATE_FLOATS + threadIdx.x;
            st[0] = T; st[GDR_BLOCK] = C0; st[2 * GDR_BLOCK] = C1; st[3 * GDR_BLOCK] = C2;
            st[4 * GDR_BLOCK] = Dp; st[5 * GDR_BLOCK] = Wt;
        }
        uint64_t live = __ballot(thr < INFINITY);
        if (lane == 0) s_done[wave] = live == 0ull ? 1 : 0;
        __syncthreads();
        if (s_done[0] + s_done[1] + s_done[2] + s_done[3] == GDR_BLOCK / GDR_WAVE) break;   // every pixel of the tile is done
        stage_write(lds, r_valid, r_xe, r_co, r_cd);
        __syncthreads();
        {   // records of the next slice (their ids arrived a round ago), ids of the one after
            r_valid = n_valid;
            if (r_valid) load_rec(n_id, r_xe, r_co, r_cd);
            const int nn = (r + 2) * GDR_BLOCK + (int)threadIdx.x;
            n_valid = nn < full_total;
            if (n_valid) n_id = point_list[range.x + (uint32_t)nn];
        }
        if (live == 0ull) continue;
        const uint32_t base = (uint32_t)pos0 + 1u;
        // this wave's row lists of the slice (compacted, see RowLists)
        int n[4] = {0, 0, 0, 0};
#if GDR_K7_STUB & 32
#pragma unroll 1
        for (int twice = 0; twice < 2; ++twice) {
        n[0] = n[1] = n[2] = n[3] = 0;
        wave_lds_fence();
#endif
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
#if GDR_K7_STUB & 32
        }
#endif
        const int nmax = max(max(n[0], n[1]), max(n[2], n[3]));
        if (nmax == 0) continue;
        wave_lds_fence();
        const uint16_t* my_list = &rlists.idx[wave][row][0];
        bool abort = false;
        auto fetch = [&](Entry& en, uint32_t e) __attribute__((always_inline)) {
            en.e = e;
            en.m = lds.xy[e]; en.co = lds.co[e]; en.cd = lds.cd[e];
        };
        auto composite = [&](const Entry& en) __attribute__((always_inline)) {
            const float dx = en.m.x - pxf, dy = en.m.y - pyf;
            const float p2 = gauss_power(dx, dy, en.co.x, en.co.y, en.co.z);
            float alpha = fminf(0.99f, en.co.w * __builtin_amdgcn_exp2f(p2));
            alpha = (p2 > 0.f) ? 0.f : alpha;        // reference skips power > 0
            if (__builtin_expect(__ballot(alpha_near_threshold(alpha)) != 0ull, 0)) {   // threshold guard (rare)
                if (alpha_near_threshold(alpha))
                    alpha = alpha_exact(rec, point_list[range.x + (uint32_t)pos0 + en.e], pxf, pyf).x;
            }
            const float a_c = (alpha >= thr) ? alpha : 0.f;
            const float T_new = fmaf(-a_c, T, T);     // T (1 - alpha); == T when a_c == 0
            const bool stop = T_new < 0.0001f;        // only possible when a_c > 0 (T >= 1e-4 while live)
            const float w = stop ? 0.f : a_c * T;
            T = stop ? T : T_new;
            thr = stop ? INFINITY : thr;
            C0 = fmaf(en.cd.x, w, C0);
            C1 = fmaf(en.cd.y, w, C1);
            C2 = fmaf(en.cd.z, w, C2);
            Dp = fmaf(en.cd.w, w, Dp);
            Wt += w;
            last_contributor = (w > 0.f) ? base + en.e : last_contributor;
            if (__ballot(stop) != 0ull) {  // rare: some pixel saturated
                live = __ballot(thr < INFINITY);
                if (live == 0ull) abort = true;
            }
        };
        // four list positions per 8-byte LDS read; entry k+1 is fetched while entry k is composited
        Entry A, B;
        uint2 q = *reinterpret_cast<const uint2*>(my_list);
        fetch(A, q.x & 0xFFFFu);
        for (int i = 0; i < nmax && !abort; i += 4) {
            const uint2 qn = *reinterpret_cast<const uint2*>(my_list + min(i + 4, GDR_BLOCK - 4));
            fetch(B, q.x >> 16);    // (read-ahead clamped to this row's own list: the last read repeats entries 252..255,
            composite(A);           //  which are never composited)
            fetch(A, q.y & 0xFFFFu);
            composite(B);
            fetch(B, q.y >> 16);
            composite(A);
            fetch(A, qn.x & 0xFFFFu);
            composite(B);
            q = qn;
        }
    }
    if (cut) {  // totals of the cut list (K7 forms "everything behind a cut" = totals - prefix)
        float* st = seg_state + ((size_t)sb + (size_t)(nseg - 1)) * GDR_SEG_STATE_FLOATS + threadIdx.x;
        st[0] = T; st[GDR_BLOCK] = C0; st[2 * GDR_BLOCK] = C1; st[3 * GDR_BLOCK] = C2;
        st[4 * GDR_BLOCK] = Dp; st[5 * GDR_BLOCK] = Wt;
    }
    if (inside) {
        const size_t pix = (size_t)py * W + px, P = (size_t)H * W;
        final_T[pix] = T;
        n_contrib[pix] = last_contributor;
        if (LOSS == 2) {   // d (mean_{c,p} (clamp(c) - target)^2) / d colour: the clamp passes the gradient inside [0,1]
            const float k = fl.go_scale * 2.f / (3.f * (float)P);
            const float c0 = fmaf(T, bg[0], C0), c1 = fmaf(T, bg[1], C1), c2 = fmaf(T, bg[2], C2);
            out_color[pix] = (c0 >= 0.f && c0 <= 1.f) ? k * (c0 - fl.target[pix]) : 0.f;
            out_color[P + pix] = (c1 >= 0.f && c1 <= 1.f) ? k * (c1 - fl.target[P + pix]) : 0.f;
            out_color[2 * P + pix] = (c2 >= 0.f && c2 <= 1.f) ? k * (c2 - fl.target[2 * P + pix]) : 0.f;
        } else {
            out_color[pix] = fmaf(T, bg[0], C0);
            out_color[P + pix] = fmaf(T, bg[1], C1);
            out_color[2 * P + pix] = fmaf(T, bg[2], C2);
            out_depth[pix] = Dp;
            out_alpha[pix] = Wt;
        }
    }
    if (LOSS) {
        float lp = 0.f;
        if (inside) {
            const size_t pix = (size_t)py * W + px, P = (size_t)H * W;
            const float invp = 1.f / (float)P;
            const float c0 = fminf(fmaxf(fmaf(T, bg[0], C0), 0.f), 1.f) - fl.target[pix];
            const float c1 = fminf(fmaxf(fmaf(T, bg[1], C1), 0.f), 1.f) - fl.target[P + pix];
            const float c2 = fminf(fmaxf(fmaf(T, bg[2], C2), 0.f), 1.f) - fl.target[2 * P + pix];
            lp = (fmaf(c0, c0, fmaf(c1, c1, c2 * c2)) * (1.f / 3.f) + fl.w_depth * Dp + fl.w_alpha * Wt) * invp;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lp += __shfl_xor(lp, off, 64);
        __syncthreads();  // s_done is free again
        float* wl = reinterpret_cast<float*>(s_done);
        if (lane == 0) wl[wave] = lp;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(fl.loss, (wl[0] + wl[1]) + (wl[2] + wl[3]));
    }
}

// ---------------------------------------------------------------------------------
// K6 "deep": the forward of CUT tile lists when few tiles are busy (seg_count[2], set by tile_order_kernel).
// An object in front of an empty background (the reference's scenes: /root/reference/dataLoader/gobjverse.py:46-104)
// puts all duplicates into a few hundred tiles with 10^4 entries each: the standard kernel then has ONE workgroup per
// CU (one wave per SIMD) walking a long list, and LDS / transcendental / DPP latencies are fully exposed (C4 `shell`:
// 897 us against 207 us for the same number of duplicates spread over 2500 tiles).  Compositing is sequential per
// pixel, so more parallelism has to come from fewer pixels per wave: here a workgroup takes an 8x8 QUARTER of the tile
// (4 workgroups per tile, each staging the tile's slices itself), a wave a 4x4 block, and the FOUR lanes of a pixel
// take four consecutive entries of the block's culled sub-list: alpha is evaluated for the four in parallel, the
// transmittance chain T_{j+1} = T_j - alpha_j T_j runs through the quad with three DPP moves (same operation order as
// the standard kernel: identical T, identical skip / stop decisions), every lane accumulates its own partial colour /
// depth / coverage sums, reduced once per cut and at the end.  A pixel that saturates inside an iteration (rare: once
// per pixel) sends its wave through a sequential replay of that iteration.  4x the waves for the same work.
// Pixel state saved at the cuts uses the standard kernel's thread order, so K7 is unchanged.
// ---------------------------------------------------------------------------------
template <int LOSS>
__global__ __launch_bounds__(GDR_BLOCK) void render_fwd_deep_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const uint32_t* __restrict__ tile_order,
    int W, int H, int gx, int ntiles, const float4* __restrict__ rec, const float* __restrict__ bg,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
    float* __restrict__ out_depth, float* __restrict__ out_alpha, const FusedLoss fl,
    const uint32_t* __restrict__ seg_base, float* __restrict__ seg_state, int seg_rounds,
    const uint32_t* __restrict__ deep_flag) {
    __shared__ SliceLds lds;
    __shared__ uint16_t clist[GDR_BLOCK / GDR_WAVE][GDR_BLOCK];
    __shared__ int s_done[GDR_BLOCK / GDR_WAVE];
    // The four quarter workgroups of a tile gather the same records: workgroups are dealt round-robin to the 8 XCDs
    // (blockIdx % 8), each with its own L2, so the quarters of one tile are blocks b, b + 8, b + 16, b + 24 — same
    // XCD, dispatched back to back — and three of the four gathers hit in L2 (HBM fetch of the kernel at C4 `shell`:
    // 1.7 GB with quarter = blockIdx & 3, which spread a tile over four XCDs).
    const uint32_t slot = (blockIdx.x >> 5) * 8u + (blockIdx.x & 7u), quarter = (blockIdx.x >> 3) & 3u;
    if (*deep_flag == 0u || (int)slot >= ntiles) return;
    const uint32_t tile = tile_order[slot];
    const uint2 range = ranges[tile];
    const int full_total = (int)(range.y - range.x);
    const int full_rounds = (full_total + GDR_BLOCK - 1) / GDR_BLOCK;
    const uint32_t sb = (seg_rounds > 0 && full_rounds > seg_rounds) ? seg_base[tile] : 0xFFFFFFFFu;
    if (sb == 0xFFFFFFFFu) return;   // not a cut tile: the standard kernel renders it
    const int nseg = (full_rounds + seg_rounds - 1) / seg_rounds;

    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t pq = lane >> 2, j = lane & 3u;   // pixel of the wave's 4x4 block, sub-lane (entry slot) of the pixel
    if (threadIdx.x == 0) {
        lds.xy[GDR_NULL_ENTRY] = make_float2(0.f, 0.f);
        lds.co[GDR_NULL_ENTRY] = make_float4(0.f, 0.f, 0.f, 0.f);
        lds.cd[GDR_NULL_ENTRY] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int tx = (int)(tile % (uint32_t)gx), ty = (int)(tile / (uint32_t)gx);
    const int bx0 = tx * GDR_TILE + (int)(quarter & 1u) * 8 + (int)(wave & 1u) * 4;
    const int by0 = ty * GDR_TILE + (int)(quarter >> 1) * 8 + (int)(wave >> 1) * 4;
    const int px = bx0 + (int)(pq & 3u), py = by0 + (int)(pq >> 2);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const float BX = (float)bx0, BY = (float)by0;
    const uint32_t tid_std = quarter * 64u + wave * 16u + pq;  // this pixel's thread in the standard kernel / in K7

    float thr = inside ? GDR_ALPHA_MIN : INFINITY;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, Wt = 0.f;   // T: the pixel's (same on its 4 lanes); sums: per lane
    uint32_t last_contributor = 0;

    auto quad_sum = [&](float v) __attribute__((always_inline)) -> float {
        v += dpp_get<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
        v += dpp_get<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
        return v;
    };
    auto save_state = [&](int slot) __attribute__((always_inline)) {
        const float c0 = quad_sum(C0), c1 = quad_sum(C1), c2 = quad_sum(C2), dp = quad_sum(Dp), wt = quad_sum(Wt);
        if (j == 0u) {
            float* st = seg_state + ((size_t)sb + (size_t)slot) * GDR_SEG_STATE_FLOATS + tid_std;
            st[0] = T; st[GDR_BLOCK] = c0; st[2 * GDR_BLOCK] = c1; st[3 * GDR_BLOCK] = c2;
            st[4 * GDR_BLOCK] = dp; st[5 * GDR_BLOCK] = wt;
        }
    };

    // one loop over the rounds of the whole list, ids two rounds and records one round ahead (see render_fwd_kernel)
    auto load_rec = [&](uint32_t id, float4& xe, float4& co, float4& cd) __attribute__((always_inline)) {
        const float4 a0 = rec[4 * (size_t)id], a3 = rec[4 * (size_t)id + 3];
        xe = make_float4(a0.x, a0.y, a3.x, a3.y); co = rec[4 * (size_t)id + 1]; cd = rec[4 * (size_t)id + 2];
    };
    float4 r_xe = make_float4(0.f, 0.f, 0.f, 0.f), r_co = r_xe, r_cd = r_xe;
    bool r_valid = (int)threadIdx.x < full_total;
    if (r_valid) load_rec(point_list[range.x + threadIdx.x], r_xe, r_co, r_cd);
    bool n_valid = GDR_BLOCK + (int)threadIdx.x < full_total;
    uint32_t n_id = n_valid ? point_list[range.x + (uint32_t)GDR_BLOCK + threadIdx.x] : 0u;
    for (int r = 0; r < full_rounds; ++r) {
        if (r > 0 && r % seg_rounds == 0) save_state(r / seg_rounds - 1);   // a cut in front of this round
        uint64_t live = __ballot(thr < INFINITY);
        if (lane == 0) s_done[wave] = live == 0ull ? 1 : 0;
        __syncthreads();
        if (s_done[0] + s_done[1] + s_done[2] + s_done[3] == GDR_BLOCK / GDR_WAVE) break;
        stage_write(lds, r_valid, r_xe, r_co, r_cd);
        __syncthreads();
        {
            r_valid = n_valid;
            if (r_valid) load_rec(n_id, r_xe, r_co, r_cd);
            const int nn = (r + 2) * GDR_BLOCK + (int)threadIdx.x;
            n_valid = nn < full_total;
            if (n_valid) n_id = point_list[range.x + (uint32_t)nn];
        }
        if (live == 0ull) continue;
        // the block's sub-list of this slice, compacted in list order into clist[wave]
        int n = 0;
#pragma unroll
        for (int g = 0; g < GDR_BLOCK / GDR_WAVE; ++g) {
            const float2 m = lds.xy[g * GDR_WAVE + (int)lane];
            const float2 hh = lds.ext[g * GDR_WAVE + (int)lane];
            const bool ov = hh.x >= 0.f && m.x + hh.x >= BX && m.x - hh.x <= BX + 3.f && m.y + hh.y >= BY &&
                            m.y - hh.y <= BY + 3.f;
            const uint64_t mk = __ballot(ov);
            const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
            if (ov) clist[wave][n + below] = (uint16_t)(g * GDR_WAVE + (int)lane);
            n += __popcll(mk);
        }
        if (n == 0) continue;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // clist is wave-private: LDS order within the wave
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint32_t base = (uint32_t)(r * GDR_BLOCK) + 1u;
        auto fetch = [&](int i, Entry& en) __attribute__((always_inline)) {
            const int idx = i + (int)j;
            en.e = idx < n ? (uint32_t)clist[wave][idx] : (uint32_t)GDR_NULL_ENTRY;
            en.m = lds.xy[en.e]; en.co = lds.co[en.e]; en.cd = lds.cd[en.e];
        };
        Entry cur, nxt;
        fetch(0, cur);
        bool all_done = false;
        for (int i = 0; i < n && !all_done; i += 4) {
            if (i + 4 < n) fetch(i + 4, nxt);
            const float dx = cur.m.x - pxf, dy = cur.m.y - pyf;
            const float p2 = gauss_power(dx, dy, cur.co.x, cur.co.y, cur.co.z);
            float alpha = fminf(0.99f, cur.co.w * __builtin_amdgcn_exp2f(p2));
            alpha = (p2 > 0.f) ? 0.f : alpha;         // (null entry: opacity 0 -> alpha 0)
            if (__builtin_expect(__ballot(alpha_near_threshold(alpha)) != 0ull, 0)) {   // threshold guard (rare)
                if (alpha_near_threshold(alpha))
                    alpha = alpha_exact(rec, point_list[range.x + (uint32_t)(r * GDR_BLOCK) + cur.e], pxf, pyf).x;
            }
            float a_c = (alpha >= thr) ? alpha : 0.f;
            // transmittance in front of this lane's entry: chain through the quad, T_{j+1} = fma(-a_j, T_j, T_j)
            float Tj = T;
            float u = fmaf(-a_c, Tj, Tj);
            float x = dpp_get<0x90, 0xf>(u);          // quad_perm [0,0,1,2]: lane j reads lane j - 1
            Tj = j >= 1u ? x : Tj;
            u = fmaf(-a_c, Tj, Tj); x = dpp_get<0x90, 0xf>(u); Tj = j >= 2u ? x : Tj;
            u = fmaf(-a_c, Tj, Tj); x = dpp_get<0x90, 0xf>(u); Tj = j >= 3u ? x : Tj;
            u = fmaf(-a_c, Tj, Tj);                   // transmittance behind this lane's entry
            float w;
            if (__ballot(u < 0.0001f) == 0ull) {      // nobody saturates in this iteration (the common case)
                w = a_c * Tj;
                T = dpp_get<0xFF, 0xf>(u);            // quad_perm [3,3,3,3]: behind the fourth entry
            } else {
                // sequential replay of the four entries, exactly the standard kernel's per-entry step
                float Tq = T, thr_q = thr;
                w = 0.f;
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) {
                    const float ak = (alpha >= thr_q) ? alpha : 0.f;
                    const float Tn = fmaf(-ak, Tq, Tq);
                    const bool stop = Tn < 0.0001f;
                    const float wk = stop ? 0.f : ak * Tq;
                    const float Ta = stop ? Tq : Tn, tha = stop ? INFINITY : thr_q;
                    w = j == k ? wk : w;
                    Tq = k == 0u ? dpp_get<0x00, 0xf>(Ta) : (k == 1u ? dpp_get<0x55, 0xf>(Ta) : (k == 2u ? dpp_get<0xAA, 0xf>(Ta) : dpp_get<0xFF, 0xf>(Ta)));
                    thr_q = k == 0u ? dpp_get<0x00, 0xf>(tha) : (k == 1u ? dpp_get<0x55, 0xf>(tha) : (k == 2u ? dpp_get<0xAA, 0xf>(tha) : dpp_get<0xFF, 0xf>(tha)));
                }
                T = Tq; thr = thr_q;
                all_done = __ballot(thr < INFINITY) == 0ull;
            }
            C0 = fmaf(cur.cd.x, w, C0);
            C1 = fmaf(cur.cd.y, w, C1);
            C2 = fmaf(cur.cd.z, w, C2);
            Dp = fmaf(cur.cd.w, w, Dp);
            Wt += w;
            last_contributor = (w > 0.f) ? base + cur.e : last_contributor;
            cur = nxt;
        }
    }
    save_state(nseg - 1);
    {
        const float c0 = quad_sum(C0), c1 = quad_sum(C1), c2 = quad_sum(C2), dp = quad_sum(Dp), wt = quad_sum(Wt);
        uint32_t lc = last_contributor;
        lc = max(lc, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lc, 0xB1, 0xf, 0xf, false));
        lc = max(lc, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lc, 0x4E, 0xf, 0xf, false));
        float lp = 0.f;
        if (inside && j == 0u) {
            const size_t pix = (size_t)py * W + px, P = (size_t)H * W;
            final_T[pix] = T;
            n_contrib[pix] = lc;
            const float f0 = fmaf(T, bg[0], c0), f1 = fmaf(T, bg[1], c1), f2 = fmaf(T, bg[2], c2);
            if (LOSS == 2) {
                const float k = fl.go_scale * 2.f / (3.f * (float)P);
                out_color[pix] = (f0 >= 0.f && f0 <= 1.f) ? k * (f0 - fl.target[pix]) : 0.f;
                out_color[P + pix] = (f1 >= 0.f && f1 <= 1.f) ? k * (f1 - fl.target[P + pix]) : 0.f;
                out_color[2 * P + pix] = (f2 >= 0.f && f2 <= 1.f) ? k * (f2 - fl.target[2 * P + pix]) : 0.f;
            } else {
                out_color[pix] = f0; out_color[P + pix] = f1; out_color[2 * P + pix] = f2;
                out_depth[pix] = dp;
                out_alpha[pix] = wt;
            }
            if (LOSS) {
                const float invp = 1.f / (float)P;
                const float e0 = fminf(fmaxf(f0, 0.f), 1.f) - fl.target[pix];
                const float e1 = fminf(fmaxf(f1, 0.f), 1.f) - fl.target[P + pix];
                const float e2 = fminf(fmaxf(f2, 0.f), 1.f) - fl.target[2 * P + pix];
                lp = (fmaf(e0, e0, fmaf(e1, e1, e2 * e2)) * (1.f / 3.f) + fl.w_depth * dp + fl.w_alpha * wt) * invp;
            }
        }
        if (LOSS) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) lp += __shfl_xor(lp, off, 64);
            __syncthreads();
            float* wl = reinterpret_cast<float*>(s_done);
            if (lane == 0) wl[wave] = lp;
            __syncthreads();
            if (threadIdx.x == 0) atomicAdd(fl.loss, (wl[0] + wl[1]) + (wl[2] + wl[3]));
        }
    }
}

// ---------------------------------------------------------------------------------
// K7.  grad_rec: (N,16) floats, one 64-byte line per Gaussian (pre-zeroed by the launcher):
//   [0..3] dL/dmean2D x, y (signed, NDC units), sum|x-term|, sum|y-term|
//   [4..6] -2 dL/dconic.x, -dL/dconic.y, -2 dL/dconic.z (sums of q dx dx, q dx dy, q dy dy; K8 scales)   [7] dL/ddepth
//   [8..10] dL/dcolour   [11] dL/dopacity
// ---------------------------------------------------------------------------------
// M2_ONLY: only dL/dmean2D (x, y, |x|, |y|) is produced, accumulated over views straight into an (N,4)
// buffer — the screen-space gradient the densification step consumes (network.py:865-878); dL/ddepth and
// dL/dalpha inputs are taken as zero (that call site differentiates an image loss only).
// LOSS: dL/dpixel computed in the prologue from the colour K6 wrote and the target (see FusedLoss) instead of read
// Everything K7 touches of ONE view.  The kernel takes a table of V <= GDR_MAX_VIEWS of them (round 4): the views of a
// multi-view node are independent given the Gaussians, and one launch over V x (segments + tiles) workgroups keeps the chip
// full across the views' tails (reference-scale scenes: a view is 1-2.5 k workgroups for 1280 resident slots) instead of V
// launches on side streams.  `order`: how the linear workgroup id maps to (view, slot) — see render_bwd_kernel.
struct BwdView {
    const uint2* ranges; const uint32_t* point_list; const uint32_t* tile_order;
    const float* bg; const float4* rec; const float* final_T; const uint32_t* n_contrib;
    const float* dL_dpix; const float* dL_ddepthpix; const float* dL_dalphapix; float* grad_rec;
    FusedLoss fl;
    const uint32_t* seg_base; const float* seg_state; const uint2* seg_extra; const uint32_t* seg_count;
    int seg_rounds, n_extra;
};
struct BwdViews { BwdView v[GDR_MAX_VIEWS]; };

// PAIRS (round 4, render_bwd_pairs_kernel).  The float atomics execute outside the L2s (TCC_EA0_ATOMIC == TCC_ATOMIC) at a
// fixed ~21 G record lines / s for the whole device, whatever a line carries (scripts/ubench/atomic_probe.hip): one line per
// (entry, 4x4 block) hit is what bounds K7 (C4: 20.5 M lines per launch = 0.98 ms of 1.10; C2 0.49 of 0.60).  Where the
// entries of a slice mostly cover BOTH blocks of a row pair (an 8x4 pixel area), the two rows walk the UNION of their
// lists in step, their totals are summed across the rows (one v_permlane16_swap) and ONE line is published for the pair —
// decided per wave and slice from the list lengths (GDR_PAIR_W lines saved per extra iteration).  Carrying the second mode
// costs the row-mode path 5-10 % (registers), so it is a second kernel: the library times both per scene shape and keeps
// the faster (api.hip, K7Tune); C2 K7 600 -> 505 us, C3 -4 %, sub-pixel Gaussians and object-like scenes stay on this one.
#define GDR_PAIR_W 6
template <bool M2_ONLY, bool LOSS, bool PAIRS>
__device__ __forceinline__ void render_bwd_body(const BwdViews& vs, int V, int interleave, int n_extra_max,
                                                int W, int H, int gx, int ntiles) {
    __shared__ SliceLds lds;
    __shared__ RowLists rlists;
    __shared__ uint32_t s_id[GDR_BLOCK + 1];

    // Per view: slots [0, n_extra_max): one segment of a cut list each (the full-length segments, i.e. the longest work
    // items, are dispatched first); slots [n_extra_max, n_extra_max + ntiles): one tile each — its whole list, or the
    // last segment of a cut list.  interleave: consecutive workgroups take the same slot of consecutive views (all
    // views' long items first: small scenes); otherwise the views follow one another (a view's records — 128 MB at 2 M
    // Gaussians — are not evicted by the other views' gathers).
    uint32_t view = 0, slot = blockIdx.x;
    if (V > 1) {
        const uint32_t per = (uint32_t)(ntiles + n_extra_max);
        if (interleave) { view = blockIdx.x % (uint32_t)V; slot = blockIdx.x / (uint32_t)V; }
        else { view = blockIdx.x / per; slot = blockIdx.x - view * per; }
    }
    const BwdView& bv = vs.v[view];
    const uint2* __restrict__ ranges = bv.ranges;
    const uint32_t* __restrict__ point_list = bv.point_list;
    const uint32_t* __restrict__ tile_order = bv.tile_order;
    const float* __restrict__ bg = bv.bg;
    const float4* __restrict__ rec = bv.rec;
    const float* __restrict__ final_T = bv.final_T;
    const uint32_t* __restrict__ n_contrib = bv.n_contrib;
    const float* __restrict__ dL_dpix = bv.dL_dpix;
    const float* __restrict__ dL_ddepthpix = bv.dL_ddepthpix;
    const float* __restrict__ dL_dalphapix = bv.dL_dalphapix;
    float* __restrict__ grad_rec = bv.grad_rec;
    const FusedLoss& fl = bv.fl;
    const uint32_t* __restrict__ seg_base = bv.seg_base;
    const float* __restrict__ seg_state = bv.seg_state;
    const uint2* __restrict__ seg_extra = bv.seg_extra;
    const uint32_t* __restrict__ seg_count = bv.seg_count;
    const int seg_rounds = bv.seg_rounds;
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
    if (seg_rounds > 0 && full_total > seg_rounds * GDR_BLOCK && seg_base[tile] != 0xFFFFFFFFu) {  // cut by K6
        const int seg_len = seg_rounds * GDR_BLOCK;
        nseg = (full_total + seg_len - 1) / seg_len;
        if (seg < 0) seg = nseg - 1;
        seg_lo = seg * seg_len;
        seg_hi = min(full_total, seg_lo + seg_len);
    }
    const int total = seg_hi - seg_lo;
    const uint32_t list_end = range.x + (uint32_t)seg_hi;
    const int rounds = (total + GDR_BLOCK - 1) / GDR_BLOCK;

    if (threadIdx.x == 0) {
        lds.xy[GDR_NULL_ENTRY] = make_float2(0.f, 0.f);
        lds.co[GDR_NULL_ENTRY] = make_float4(0.f, 0.f, 0.f, 0.f);
        lds.cd[GDR_NULL_ENTRY] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_id[GDR_NULL_ENTRY] = 0;
    }
    if (threadIdx.x < 8) rlists.pad[threadIdx.x] = (uint16_t)GDR_NULL_ENTRY;
    const float T_final = inside ? final_T[pix] : 0.f;
    float T = T_final;
    const int lc_full = inside ? (int)n_contrib[pix] : 0;
    // positions are counted from seg_lo below; a pixel whose last contributor lies behind this segment starts
    // from the state K6 saved at the cut (see the B initialisation further down)
    const bool from_cut = lc_full > seg_hi;
    const int last_contributor = from_cut ? total : max(lc_full - seg_lo, 0);
    float gC0 = 0.f, gC1 = 0.f, gC2 = 0.f, gD = 0.f, gA = 0.f;
    if (LOSS) {
        if (inside && last_contributor > 0) {
            const float go = *fl.go, invp = go / (float)P, k = 2.f * invp * (1.f / 3.f);
            const float c0 = fl.color[pix], c1 = fl.color[P + pix], c2 = fl.color[2 * P + pix];
            gC0 = (c0 >= 0.f && c0 <= 1.f) ? k * (c0 - fl.target[pix]) : 0.f;   // clamp passes the gradient inside [0,1]
            gC1 = (c1 >= 0.f && c1 <= 1.f) ? k * (c1 - fl.target[P + pix]) : 0.f;
            gC2 = (c2 >= 0.f && c2 <= 1.f) ? k * (c2 - fl.target[2 * P + pix]) : 0.f;
            gD = fl.w_depth * invp;
            gA = fl.w_alpha * invp;
        }
    } else if (inside && last_contributor > 0) {  // a pixel without contributors reads no upstream gradient (NaN-safe)
        gC0 = dL_dpix[pix]; gC1 = dL_dpix[P + pix]; gC2 = dL_dpix[2 * P + pix];
        if (!M2_ONLY && dL_ddepthpix) gD = dL_ddepthpix[pix];
        if (!M2_ONLY && dL_dalphapix) gA = dL_dalphapix[pix];
    }
    const float bgT = -T_final * ((bg[0] * gC0 + bg[1] * gC1) + bg[2] * gC2);
    // "behind" state B = colour / depth / coverage composited from everything behind the
    // current Gaussian; update B <- B + a (c - B) is the identity for a = 0
    float B0 = 0.f, B1 = 0.f, B2 = 0.f, BD = 0.f, BA = 0.f;
    if (from_cut) {  // (only in workgroups of seg_extra): T in front of position seg_hi, B = (totals - prefix) / T
        const float* cu = seg_state + ((size_t)seg_base[tile] + (size_t)seg) * GDR_SEG_STATE_FLOATS + threadIdx.x;
        const float* to = seg_state + ((size_t)seg_base[tile] + (size_t)(nseg - 1)) * GDR_SEG_STATE_FLOATS + threadIdx.x;
        T = cu[0];
        const float rT = 1.f / T;
        B0 = (to[GDR_BLOCK] - cu[GDR_BLOCK]) * rT;
        B1 = (to[2 * GDR_BLOCK] - cu[2 * GDR_BLOCK]) * rT;
        B2 = (to[3 * GDR_BLOCK] - cu[3 * GDR_BLOCK]) * rT;
        BD = (to[4 * GDR_BLOCK] - cu[4 * GDR_BLOCK]) * rT;
        BA = (to[5 * GDR_BLOCK] - cu[5 * GDR_BLOCK]) * rT;
    }
    // the staged conic is log2(e) x the true one: fold 1/log2(e) = ln 2 into the pixel->NDC factors
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

    // slice r holds list positions total-1-(r*256+e), e = 0..255: back to front
    float4 r_xe = make_float4(0.f, 0.f, 0.f, 0.f), r_co = r_xe, r_cd = r_xe;
    uint32_t r_id = 0;
    bool r_valid = (int)threadIdx.x < total;
    if (r_valid) {
        r_id = point_list[list_end - 1u - threadIdx.x];
        { const float4 a0 = rec[4 * (size_t)r_id], a3 = rec[4 * (size_t)r_id + 3];
          r_xe = make_float4(a0.x, a0.y, a3.x, a3.y); r_co = rec[4 * (size_t)r_id + 1]; r_cd = rec[4 * (size_t)r_id + 2]; }
    }
    for (int r = 0; r < rounds; ++r) {
        __syncthreads();  // every wave finished reading the previous slice
        stage_write(lds, r_valid, r_xe, r_co, r_cd);
        if (r_valid) s_id[threadIdx.x] = r_id;
        __syncthreads();
        {
            const int nxt = (r + 1) * GDR_BLOCK + (int)threadIdx.x;
            r_valid = nxt < total;
            if (r_valid) {
                r_id = point_list[list_end - 1u - (uint32_t)nxt];
                { const float4 a0 = rec[4 * (size_t)r_id], a3 = rec[4 * (size_t)r_id + 3];
          r_xe = make_float4(a0.x, a0.y, a3.x, a3.y); r_co = rec[4 * (size_t)r_id + 1]; r_cd = rec[4 * (size_t)r_id + 2]; }
            }
        }
        const int top = total - 1 - r * GDR_BLOCK;  // list position of LDS entry e: top - e
        if (top - (GDR_BLOCK - 1) >= wave_last) continue;  // whole slice behind every last contributor
        // Row sub-lists of the slice's four 64-entry groups, one 64-bit mask per group and lane (all 16 lanes of a row
        // hold the same four masks).  A row walks them back to back at its own pace — rows only re-synchronise at slice
        // boundaries, so a row whose block has few entries in one group does not wait for the others there.
        // this wave's compacted row lists of the slice (RowLists): entries in front of each block's deepest contributor
        int n[4] = {0, 0, 0, 0}, un[2] = {0, 0};
#if GDR_K7_STUB & 8
#pragma unroll 1
        for (int twice = 0; twice < 2; ++twice) {
        n[0] = n[1] = n[2] = n[3] = 0; un[0] = un[1] = 0;
        wave_lds_fence();
#endif
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
#if GDR_K7_STUB & 8
        }
#endif
        int nmax = max(max(n[0], n[1]), max(n[2], n[3]));
        if (nmax == 0) continue;
        bool pair_mode = false;
        if (PAIRS) {
            const int umax = max(un[0], un[1]);
            pair_mode = (umax - nmax) * GDR_PAIR_W < (n[0] + n[1] + n[2] + n[3]) - (un[0] + un[1]);
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
            auto fetch = [&](Entry& en, uint32_t e) __attribute__((always_inline)) {
                en.e = e;
                en.m = lds.xy[e]; en.co = lds.co[e]; en.cd = lds.cd[e];
            };
            auto accumulate = [&](const Entry& en) {
                const float dx = en.m.x - pxf, dy = en.m.y - pyf;
                const float p2 = gauss_power(dx, dy, en.co.x, en.co.y, en.co.z);
                float G = __builtin_amdgcn_exp2f(p2);
                float alpha = fminf(0.99f, en.co.w * G);
                alpha = (p2 > 0.f) ? 0.f : alpha;
                if (__builtin_expect(__ballot(alpha_near_threshold(alpha)) != 0ull, 0)) {   // threshold guard (rare): as K6
                    if (alpha_near_threshold(alpha)) {
                        const float2 ag = alpha_exact(rec, s_id[en.e], pxf, pyf);
                        alpha = ag.x; G = ag.y;
                    }
                }
                // contributes iff it did in the forward: alpha >= 1/255 and position < last_contributor
                // (the null entry has opacity 0 -> alpha 0)
                const float lim = (top - (int)en.e < last_contributor) ? GDR_ALPHA_MIN : INFINITY;
                const bool hit = alpha >= lim;
                const uint64_t hb = __ballot(hit);
#ifndef GDR_K7_NO_EARLYOUT      /* measurement builds: without the branch the four accumulates of the unrolled walk are one basic block */
                if (hb == 0ull) return;
#endif
                const float a = hit ? alpha : 0.f;
                const float Gh = hit ? G : 0.f;                       // a lane without a hit contributes nothing below
                const float r_oma = __builtin_amdgcn_rcpf(1.f - a);  // == 1 when a == 0
                T = T * r_oma;                                        // transmittance in FRONT of this Gaussian
                const float w = a * T;
                const float d0 = en.cd.x - B0, d1 = en.cd.y - B1, d2 = en.cd.z - B2;
                float dL_dalpha;
#if GDR_K7_STUB & 16
                dL_dalpha = T + d0 + d1 + d2;
#else
                if (M2_ONLY) {
                    dL_dalpha = fmaf(d0, gC0, fmaf(d1, gC1, d2 * gC2));
                } else {
                    const float dD = en.cd.w - BD, dA = 1.f - BA;
                    dL_dalpha = fmaf(d0, gC0, fmaf(d1, gC1, fmaf(d2, gC2, fmaf(dD, gD, dA * gA))));
                    BD = fmaf(a, dD, BD); BA = fmaf(a, dA, BA);
                }
                dL_dalpha = fmaf(dL_dalpha, T, bgT * r_oma);
                B0 = fmaf(a, d0, B0); B1 = fmaf(a, d1, B1); B2 = fmaf(a, d2, B2);
#endif
                // go = G dL/dalpha (the opacity term), q = opacity * go = G dL/dG; everything below is q times a polynomial in
                // (dx, dy): formed from q dx and q dy, 6 + 6 multiplies for the five geometric terms instead of 11 + 8
                const float go = Gh * dL_dalpha;
                const float q = en.co.w * go;
                const float qdx = q * dx, qdy = q * dy;
                // co.xyz carry the log2(e) factor; kx, ky carry its inverse
                const float v_mx = fmaf(qdx, en.co.x, qdy * en.co.y) * -kx;
                const float v_my = fmaf(qdy, en.co.z, qdx * en.co.y) * -ky;
                const bool publish = PAIRS ? (hb & unit_mask) != 0ull : ((hb >> (16 * row)) & 0xFFFFull) != 0ull;
                if (M2_ONLY) {
                    float tot4 = row_reduce_scatter4(v_mx, v_my, fabsf(v_mx), fabsf(v_my), li);
                    if (PAIRS && pair_mode) tot4 = rows2_sum(tot4);
                    if ((li & 3u) == 0u && publish) atomicAdd(grad_rec + 4 * (size_t)s_id[en.e] + (li >> 2), tot4);
                    return;
                }
                // (record words 4..6 = sum of q dx dx, q dx dy, q dy dy: K8 applies the exact factors -1/2, -1, -1/2)
#if GDR_K7_STUB & 4
                const float vals[12] = {dL_dalpha, w, dL_dalpha, w, dL_dalpha, w, dL_dalpha, w, dL_dalpha, w, dL_dalpha, w};
#else
                const float vals[12] = {v_mx, v_my, fabsf(v_mx), fabsf(v_my), qdx * dx, qdx * dy, qdy * dy,
                                        w * gD, w * gC0, w * gC1, w * gC2, go};
#endif
#if GDR_K7_STUB & 2
                float tot = (((vals[0] + vals[1]) + (vals[2] + vals[3])) + ((vals[4] + vals[5]) + (vals[6] + vals[7]))) +
                            ((vals[8] + vals[9]) + (vals[10] + vals[11]));
#else
                float tot = row_reduce_scatter12(vals, li);
#endif
                if (PAIRS && pair_mode) tot = rows2_sum(tot);
                // lanes 0..11 of every row (pair mode: row pair) that had a hit add the totals to the Gaussian's
                // 64-byte gradient record: one global_atomic_add_f32 instruction, one cache line
                // per row (no return value => fire and forget)
#if GDR_K7_STUB & 1
                if (li < 12u && publish && tot == 12345.678f)
#else
                if (li < 12u && publish)
#endif
                    atomicAdd(grad_rec + 16 * (size_t)s_id[en.e] + li, tot);
            };
            // four list positions per 8-byte LDS read; entry k+1 is fetched while entry k is accumulated
            Entry A, B;
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

template <bool M2_ONLY, bool LOSS = false>
__global__ __launch_bounds__(GDR_BLOCK) GDR_WALK_WAVES GDR_K7_OCC
void render_bwd_kernel(const BwdViews vs, int V, int interleave, int n_extra_max,
                                                               int W, int H, int gx, int ntiles) {
    render_bwd_body<M2_ONLY, LOSS, false>(vs, V, interleave, n_extra_max, W, H, gx, ntiles);
}
// (five waves per SIMD as the row-mode kernel: the handful of registers beyond 96 are spilled outside the walk)
template <bool M2_ONLY, bool LOSS = false>
__global__ __launch_bounds__(GDR_BLOCK) __attribute__((amdgpu_waves_per_eu(5, 5)))
void render_bwd_pairs_kernel(const BwdViews vs, int V, int interleave, int n_extra_max, int W, int H, int gx, int ntiles) {
    render_bwd_body<M2_ONLY, LOSS, true>(vs, V, interleave, n_extra_max, W, H, gx, ntiles);
}

}  // namespace

// the K7 variant of the calling thread's next launches (api.hip: K7Scope)
static thread_local int t_bwd_pairs = 0;
void render_bwd_set_pairs(int pairs) { t_bwd_pairs = pairs; }

hipError_t launch_tile_order_views(const BinViews& vs, int V, int ntiles, hipStream_t st) {
    GDR_LAUNCH(GDR_K_TILE_ORDER, tile_order_kernel, dim3(1, V), dim3(GDR_ORDER_THREADS), st, vs, ntiles);
    return hipGetLastError();
}


// cut tiles in "deep" mode (seg_count[2]): 4 workgroups per cut tile, in front of the standard launch
#define GDR_DEEP_FLAG(bin, img) (seg_rounds_of(bin, img) && !(bin)->hint_no_deep ? (const uint32_t*)(bin)->seg_count + 2 : nullptr)
#define GDR_DEEP_LAUNCH(LOSSV, COLOR, DEPTH, ALPHA, FL)                                                                  \
    do {                                                                                                                  \
        if (seg_rounds_of(bin, img) && img->tile_order && bin->deep_max_busy > 0 && !bin->hint_no_deep) {                                    \
            int ndeep = bin->seg_cap < ntiles ? bin->seg_cap : ntiles;     /* cut tiles <= busy tiles <= deep_max_busy */ \
            ndeep = 4 * (((ndeep < bin->deep_max_busy ? ndeep : bin->deep_max_busy) + 7) / 8 * 8);                        \
            GDR_LAUNCH(GDR_K_RENDER_FWD_DEEP, render_fwd_deep_kernel<LOSSV>, dim3(ndeep), dim3(GDR_BLOCK), st,             \
                       (const uint2*)img->ranges, bin->values[bin->sorted], img->tile_order, W, H, gx, ntiles,            \
                       (const float4*)g->rec, s->bg, img->final_T, img->n_contrib, COLOR, DEPTH, ALPHA, FL,               \
                       GDR_SEG_FWD_ARGS(bin, img), GDR_DEEP_FLAG(bin, img));                                              \
        }                                                                                                                 \
    } while (0)

// ---- K6 launchers: the deep launch of every view that may need one, then ONE table-driven standard launch ----
namespace {
struct FwdSpec {   // host side of FwdView
    const gdr_settings* s; const gdr_geom* g; const gdr_binning* bin; const gdr_image* img;
    float* color; float* depth; float* alpha;
    FusedLoss fl;
};

template <int LOSS>
hipError_t launch_fwd_table(int V, const FwdSpec* sp, int interleave, hipStream_t st) {
    const int W = sp[0].s->image_width, H = sp[0].s->image_height;
    const int gx = tile_grid_x(W), gy = tile_grid_y(H);
    const int ntiles = gx * gy;
    FwdViews vs;
    memset(&vs, 0, sizeof(vs));
    for (int v = 0; v < V; ++v) {
        const gdr_settings* s = sp[v].s; const gdr_geom* g = sp[v].g; const gdr_binning* bin = sp[v].bin; const gdr_image* img = sp[v].img;
        GDR_DEEP_LAUNCH(LOSS, sp[v].color, sp[v].depth, sp[v].alpha, sp[v].fl);
        FwdView& f = vs.v[v];
        f.ranges = (const uint2*)img->ranges; f.point_list = bin->values[bin->sorted]; f.tile_order = img->tile_order;
        f.rec = (const float4*)g->rec; f.bg = s->bg; f.final_T = img->final_T; f.n_contrib = img->n_contrib;
        f.out_color = sp[v].color; f.out_depth = sp[v].depth; f.out_alpha = sp[v].alpha; f.fl = sp[v].fl;
        f.seg_base = img->seg_base; f.seg_state = bin->seg_state; f.seg_rounds = seg_rounds_of(bin, img);
        f.deep_flag = img->tile_order ? GDR_DEEP_FLAG(bin, img) : nullptr;
    }
    GDR_LAUNCH(GDR_K_RENDER_FWD, render_fwd_kernel<LOSS>, dim3((unsigned)V * (unsigned)ntiles), dim3(GDR_BLOCK), st, vs, V, interleave,
               W, H, gx, ntiles);
    return hipGetLastError();
}
}  // namespace

hipError_t launch_render_fwd(const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                             const gdr_image* img, const gdr_outputs* out, hipStream_t st) {
    const FwdSpec sp{s, g, bin, img, out->color, out->depth, out->alpha, FusedLoss{}};
    return launch_fwd_table<0>(1, &sp, 0, st);
}

hipError_t launch_render_fwd_loss(const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                  const gdr_image* img, const gdr_outputs* out, const float* target, float w_depth,
                                  float w_alpha, float* loss, hipStream_t st) {
    const FwdSpec sp{s, g, bin, img, out->color, out->depth, out->alpha, FusedLoss{target, w_depth, w_alpha, loss, nullptr, nullptr, 1.f}};
    return launch_fwd_table<1>(1, &sp, 0, st);
}

// K6 of the abs-grad-only path: loss accumulated, d loss / d colour written instead of any image (LOSS = 2)
hipError_t launch_render_fwd_lossgrad(const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                      const gdr_image* img, const float* target, float go_scale, float* loss,
                                      float* dL_dcolor, hipStream_t st) {
    const FwdSpec sp{s, g, bin, img, dL_dcolor, nullptr, nullptr, FusedLoss{target, 0.f, 0.f, loss, nullptr, nullptr, go_scale}};
    return launch_fwd_table<2>(1, &sp, 0, st);
}

// K6 of V <= GDR_MAX_VIEWS views of one image size in ONE launch (gdr_composite_forward_views).  loss_mode 0: images only;
// 1: images + the folded loss (losses[v] accumulated); 2: loss + d loss / d colour into outs[v].color, no image
hipError_t launch_render_fwd_views(int V, const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin, const gdr_image* img,
                                   const gdr_outputs* outs, int loss_mode, const float* const* targets, float w_depth,
                                   float w_alpha, float go_scale, float* losses, int interleave, hipStream_t st) {
    FwdSpec sp[GDR_MAX_VIEWS];
    for (int v = 0; v < V; ++v) {
        FusedLoss fl{};
        if (loss_mode == 1) fl = FusedLoss{targets[v], w_depth, w_alpha, losses + v, nullptr, nullptr, 1.f};
        if (loss_mode == 2) fl = FusedLoss{targets[v], 0.f, 0.f, losses + v, nullptr, nullptr, go_scale};
        sp[v] = FwdSpec{&s[v], &g[v], &bin[v], &img[v], outs[v].color, loss_mode == 2 ? nullptr : outs[v].depth,
                        loss_mode == 2 ? nullptr : outs[v].alpha, fl};
    }
    if (loss_mode == 1) return launch_fwd_table<1>(V, sp, interleave, st);
    if (loss_mode == 2) return launch_fwd_table<2>(V, sp, interleave, st);
    return launch_fwd_table<0>(V, sp, interleave, st);
}

// ---- K7 launchers: every variant goes through ONE table-driven launch (V = 1 for the single-view entry points) ----
namespace {
struct BwdSpec {   // host side of BwdView
    const gdr_settings* s; const gdr_geom* g; const gdr_binning* bin; const gdr_image* img;
    const float* dL_dcolor; const float* dL_ddepth; const float* dL_dalpha;
    FusedLoss fl;
    float* out;      // (N,16) gradient records, or the (N,4) mean2D buffer of the M2_ONLY variants
};

template <bool M2_ONLY, bool LOSS>
hipError_t launch_bwd_table(int V, const BwdSpec* sp, int interleave, hipStream_t st) {
    const int W = sp[0].s->image_width, H = sp[0].s->image_height;
    const int gx = tile_grid_x(W), gy = tile_grid_y(H);
    const int ntiles = gx * gy;
    BwdViews vs;
    memset(&vs, 0, sizeof(vs));
    int n_extra_max = 0;
    for (int v = 0; v < V; ++v) {
        const BwdSpec& p = sp[v];
        BwdView& b = vs.v[v];
        b.ranges = (const uint2*)p.img->ranges; b.point_list = p.bin->values[p.bin->sorted]; b.tile_order = p.img->tile_order;
        b.bg = p.s->bg; b.rec = (const float4*)p.g->rec; b.final_T = p.img->final_T; b.n_contrib = p.img->n_contrib;
        b.dL_dpix = p.dL_dcolor; b.dL_ddepthpix = p.dL_ddepth; b.dL_dalphapix = p.dL_dalpha; b.grad_rec = p.out;
        b.fl = p.fl;
        b.seg_rounds = seg_rounds_of(p.bin, p.img);
        b.n_extra = b.seg_rounds ? p.bin->seg_cap : 0;
        b.seg_base = p.img->seg_base; b.seg_state = (const float*)p.bin->seg_state;
        b.seg_extra = (const uint2*)p.bin->seg_extra; b.seg_count = p.bin->seg_count;
        n_extra_max = b.n_extra > n_extra_max ? b.n_extra : n_extra_max;
    }
    const unsigned grid = (unsigned)V * (unsigned)(ntiles + n_extra_max);
    if (t_bwd_pairs)
        GDR_LAUNCH(GDR_K_RENDER_BWD, (render_bwd_pairs_kernel<M2_ONLY, LOSS>), dim3(grid), dim3(GDR_BLOCK), st, vs, V, interleave,
                   n_extra_max, W, H, gx, ntiles);
    else
        GDR_LAUNCH(GDR_K_RENDER_BWD, (render_bwd_kernel<M2_ONLY, LOSS>), dim3(grid), dim3(GDR_BLOCK), st, vs, V, interleave,
                   n_extra_max, W, H, gx, ntiles);
    return hipGetLastError();
}
}  // namespace

hipError_t launch_render_bwd_loss(const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                  const gdr_image* img, const float* color, const float* target, float w_depth,
                                  float w_alpha, const float* go, float* grad_rec, hipStream_t st) {
    const BwdSpec sp{s, g, bin, img, nullptr, nullptr, nullptr, FusedLoss{target, w_depth, w_alpha, nullptr, go, color, 1.f}, grad_rec};
    return launch_bwd_table<false, true>(1, &sp, 0, st);
}

hipError_t launch_render_bwd(const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                             const gdr_image* img, const gdr_grad_inputs* gi,
                             const gdr_grad_outputs* go, hipStream_t st) {
    const BwdSpec sp{s, g, bin, img, gi->dL_dcolor, gi->dL_ddepth, gi->dL_dalpha, FusedLoss{}, go->scratch};
    return launch_bwd_table<false, false>(1, &sp, 0, st);
}

// K7 of V <= GDR_MAX_VIEWS views of one image size in ONE launch (gdr_render_backward_views / _loss_views)
hipError_t launch_render_bwd_views(int V, const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                   const gdr_image* img, const gdr_grad_inputs* gi, float* const* grad_recs,
                                   int interleave, hipStream_t st) {
    BwdSpec sp[GDR_MAX_VIEWS];
    for (int v = 0; v < V; ++v)
        sp[v] = BwdSpec{&s[v], &g[v], &bin[v], &img[v], gi[v].dL_dcolor, gi[v].dL_ddepth, gi[v].dL_dalpha, FusedLoss{}, grad_recs[v]};
    return launch_bwd_table<false, false>(V, sp, interleave, st);
}

hipError_t launch_render_bwd_loss_views(int V, const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                        const gdr_image* img, const float* const* colors, const float* const* targets,
                                        float w_depth, float w_alpha, const float* go, float* const* grad_recs,
                                        int interleave, hipStream_t st) {
    BwdSpec sp[GDR_MAX_VIEWS];
    for (int v = 0; v < V; ++v)
        sp[v] = BwdSpec{&s[v], &g[v], &bin[v], &img[v], nullptr, nullptr, nullptr,
                        FusedLoss{targets[v], w_depth, w_alpha, nullptr, go + v, colors[v], 1.f}, grad_recs[v]};
    return launch_bwd_table<false, true>(V, sp, interleave, st);
}

hipError_t launch_render_bwd_mean2d(const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                    const gdr_image* img, const float* dL_dcolor, float* dL_dmean2D,
                                    hipStream_t st) {
    const BwdSpec sp{s, g, bin, img, dL_dcolor, nullptr, nullptr, FusedLoss{}, dL_dmean2D};
    return launch_bwd_table<true, false>(1, &sp, 0, st);
}

// abs-grad-only K7 with the MSE loss folded into its prologue (network.py:865-878 differentiates an image
// MSE w.r.t. the means2D carrier only)
hipError_t launch_render_bwd_mean2d_loss(const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                         const gdr_image* img, const float* color, const float* target, const float* go,
                                         float* dL_dmean2D, hipStream_t st) {
    const BwdSpec sp{s, g, bin, img, nullptr, nullptr, nullptr, FusedLoss{target, 0.f, 0.f, nullptr, go, color, 1.f}, dL_dmean2D};
    return launch_bwd_table<true, true>(1, &sp, 0, st);
}

// the abs-grad-only K7 of V views accumulating into ONE (N,4) buffer (gdr_render_backward_mean2d_views)
hipError_t launch_render_bwd_mean2d_views(int V, const gdr_settings* s, const gdr_geom* g, const gdr_binning* bin,
                                          const gdr_image* img, const float* const* dL_dcolors, float* dL_dmean2D,
                                          int interleave, hipStream_t st) {
    BwdSpec sp[GDR_MAX_VIEWS];
    for (int v = 0; v < V; ++v)
        sp[v] = BwdSpec{&s[v], &g[v], &bin[v], &img[v], dL_dcolors[v], nullptr, nullptr, FusedLoss{}, dL_dmean2D};
    return launch_bwd_table<true, false>(V, sp, interleave, st);
}

}  // namespace gdr
