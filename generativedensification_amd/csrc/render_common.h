// render_common.h — wave-level helpers shared by render.hip (3DGS K6/K7) and render_surfel.hip (2DGS K6s/K7s):
// XCD-aware tile mapping, DPP row exchanges, the row-local reduce-scatters that put one per-Gaussian total on
// each lane of a 16-lane row, and the per-lane sub-list mask walk.  gfx950 only (wave64, 16-lane DPP rows).
#pragma once
#include "gdr_common.h"

namespace gdr {
namespace {

#define GDR_LOG2E 1.4426950408889634f
#define GDR_LN2 0.6931471805599453f
#define GDR_ALPHA_MIN (1.f / 255.f)
#define GDR_NULL_ENTRY GDR_BLOCK  // LDS slot 256: xy = 0, conic = 0, opacity = 0, colour = 0

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// XCD-aware bijective remap of a linear workgroup id (8 XCDs, round-robin dispatch).
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n) {
    const uint32_t xcd = b & 7u, q = n >> 3, r = n & 7u;
    const uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (b >> 3);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_get(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}

// Row-local reduce-scatter of 12 (16) per-lane values over the 16 lanes of a DPP row: lane i of the row returns the
// row total of value i (lanes 12..15 of the 12-value form return unused sums).  Four halving exchanges: row_mirror
// (partner 15 - i), row_half_mirror (i ^ 7), quad reverse (i ^ 3), quad xor-1 — the totals land one per lane, so ONE
// atomic instruction publishes all of them.  In the first two exchanges the lanes that keep the upper value of a
// pair are whole 4-lane DPP banks, so each output is two bank-masked v_add_f32_dpp into the same register
//     a = a + perm(a)   [banks of the lower half, in place]      a = b + perm(b)   [banks of the upper half]
// (all in place: no temporaries) instead of two v_cndmask + one add; the compiler has no intrinsic for a bank-masked DPP add (update_dpp with a
// partial bank mask is a v_mov and is not folded), hence the inline asm (29 VALU ops for 12 values instead of 47;
// K7 381 -> see DESIGN.md).  The last two exchanges split inside a bank and stay select + add.  s_nop 1 on both
// sides: the hazard recogniser does not look inside inline asm (VALU write -> DPP read needs two wait states).
__device__ __forceinline__ float row_reduce_scatter12(const float (&v)[12], uint32_t li) {
    const bool b1 = li & 2u, b0 = li & 1u;
    float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3], a4 = v[4], a5 = v[5], a6 = v[6], a7 = v[7], s2[2];
    asm(
        "s_nop 1\n\t"  /* VALU write -> DPP read of the inputs needs 2 wait states */
        "v_add_f32_dpp %[a0], %[a0], %[a0] row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %[a0], %[b0], %[b0] row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[a1], %[a1], %[a1] row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %[a1], %[b1], %[b1] row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[a2], %[a2], %[a2] row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %[a2], %[b2], %[b2] row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[a3], %[a3], %[a3] row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %[a3], %[b3], %[b3] row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[a4], %[a4], %[a4] row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[a5], %[a5], %[a5] row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[a6], %[a6], %[a6] row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[a7], %[a7], %[a7] row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[a0], %[a0], %[a0] row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %[a0], %[a4], %[a4] row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %[a1], %[a1], %[a1] row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %[a1], %[a5], %[a5] row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %[a2], %[a2], %[a2] row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %[a2], %[a6], %[a6] row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %[a3], %[a3], %[a3] row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %[a3], %[a7], %[a7] row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 1"      /* ... and so does the compiler-generated DPP read of a0..a3 that follows */
        : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6), [a7] "+v"(a7)
        : [b0] "v"(v[8]), [b1] "v"(v[9]), [b2] "v"(v[10]), [b3] "v"(v[11]));
    const float t[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float keep = b1 ? t[j + 2] : t[j], send = b1 ? t[j] : t[j + 2];
        s2[j] = keep + dpp_get<0x1B, 0xf>(send);  // quad_perm [3,2,1,0]: partner i ^ 3
    }
    const float keep = b0 ? s2[1] : s2[0], send = b0 ? s2[0] : s2[1];
    return keep + dpp_get<0xB1, 0xf>(send);       // quad_perm [1,0,3,2]: partner i ^ 1
}

__device__ __forceinline__ float row_reduce_scatter16(const float (&v)[16], uint32_t li) {
    const bool b1 = li & 2u, b0 = li & 1u;
    float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3], a4 = v[4], a5 = v[5], a6 = v[6], a7 = v[7], s2[2];
    asm(
        "s_nop 1\n\t"  /* VALU write -> DPP read of the inputs needs 2 wait states */
        "v_add_f32_dpp %[a0], %[a0], %[a0] row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %[a0], %[b0], %[b0] row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[a1], %[a1], %[a1] row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %[a1], %[b1], %[b1] row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[a2], %[a2], %[a2] row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %[a2], %[b2], %[b2] row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[a3], %[a3], %[a3] row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %[a3], %[b3], %[b3] row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[a4], %[a4], %[a4] row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %[a4], %[b4], %[b4] row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[a5], %[a5], %[a5] row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %[a5], %[b5], %[b5] row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[a6], %[a6], %[a6] row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %[a6], %[b6], %[b6] row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[a7], %[a7], %[a7] row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %[a7], %[b7], %[b7] row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[a0], %[a0], %[a0] row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %[a0], %[a4], %[a4] row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %[a1], %[a1], %[a1] row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %[a1], %[a5], %[a5] row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %[a2], %[a2], %[a2] row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %[a2], %[a6], %[a6] row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %[a3], %[a3], %[a3] row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %[a3], %[a7], %[a7] row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 1"      /* ... and so does the compiler-generated DPP read of a0..a3 that follows */
        : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6), [a7] "+v"(a7)
        : [b0] "v"(v[8]), [b1] "v"(v[9]), [b2] "v"(v[10]), [b3] "v"(v[11]), [b4] "v"(v[12]), [b5] "v"(v[13]), [b6] "v"(v[14]), [b7] "v"(v[15]));
    const float t[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float keep = b1 ? t[j + 2] : t[j], send = b1 ? t[j] : t[j + 2];
        s2[j] = keep + dpp_get<0x1B, 0xf>(send);  // quad_perm [3,2,1,0]: partner i ^ 3
    }
    const float keep = b0 ? s2[1] : s2[0], send = b0 ? s2[0] : s2[1];
    return keep + dpp_get<0xB1, 0xf>(send);       // quad_perm [1,0,3,2]: partner i ^ 1
}

// x of this DPP row + x of the other row of its pair (rows 0/1, 2/3), on both: v_permlane16_swap (gfx950) exchanges the
// odd rows of its first operand with the even rows of the second
__device__ __forceinline__ float rows2_sum(float x) {
    const uint32_t u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// 4 values over the 16 lanes of a row: lane i returns the row total of value (i >> 2) & 3
// (two halving exchanges, then two plain butterfly adds inside the quads)
__device__ __forceinline__ float row_reduce_scatter4(float v0, float v1, float v2, float v3, uint32_t li) {
    const bool b3 = li & 8u, b2 = li & 4u;
    const float k0 = b3 ? v2 : v0, s0 = b3 ? v0 : v2, k1 = b3 ? v3 : v1, s1 = b3 ? v1 : v3;
    const float u0 = k0 + dpp_get<0x140, 0xf>(s0), u1 = k1 + dpp_get<0x140, 0xf>(s1);  // row_mirror
    const float k = b2 ? u1 : u0, sd = b2 ? u0 : u1;
    float t = k + dpp_get<0x141, 0xf>(sd);  // row_half_mirror
    t += dpp_get<0x1B, 0xf>(t);             // quad reverse
    t += dpp_get<0xB1, 0xf>(t);             // quad xor-1  -> all 4 lanes of the quad hold the total
    return t;
}

// row total of one value, on every lane of the row
__device__ __forceinline__ float row_sum(float x) {
    x += dpp_get<0x140, 0xf>(x);   // row_mirror
    x += dpp_get<0x141, 0xf>(x);   // row_half_mirror
    x += dpp_get<0x1B, 0xf>(x);    // quad reverse
    x += dpp_get<0xB1, 0xf>(x);    // quad xor-1
    return x;
}

// Slice-wide row lists.  A 256-entry slice is culled in four 64-entry groups; every lane keeps its row's four 64-bit
// sub-list masks (q0..q3) and walks them back to back: `mr` = the rest of the current group's mask, `gi` = its index.
// Rows only re-synchronise at slice boundaries, so a block with few entries in one group does not wait for the other
// three blocks there (measured: K7 409 -> 381 us at C4, 196 -> 172 us at C2).  GDR_REFILL steps a lane over exhausted
// groups; the masks are copied to prvalues first — a `c ? q1 : q2` on by-reference lambda captures is an ADDRESS
// select, which the optimiser turns into an indexed load from a closure kept in scratch memory.
#define GDR_REFILL(mr, gi, q1, q2, q3)                                           \
    do {                                                                          \
        _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_) {                        \
            const bool adv_ = (mr) == 0ull && (gi) < 3u;                          \
            const uint64_t a1_ = (q1), a2_ = (q2), a3_ = (q3);                    \
            uint64_t nxt_ = (gi) == 0u ? a1_ : a2_;                               \
            nxt_ = (gi) >= 2u ? a3_ : nxt_;                                       \
            (mr) = adv_ ? nxt_ : (mr);                                            \
            (gi) = adv_ ? (gi) + 1u : (gi);                                       \
        }                                                                         \
    } while (0)

#define GDR_ROW_MASK(k) (0xFFFFull << (16 * (k)))

// Per-LANE mask walk: every lane carries its row's 64-bit sub-list mask in two VGPRs and takes
// the first set bit (0xFFFFFFFF if none) with VALU ops (v_ffbl_b32 x2).  The scalar unit is
// shared by the CU's four SIMDs; walking four SGPR masks cost ~25 SALU per iteration.
__device__ __forceinline__ uint32_t take_bit(uint64_t& m) {
    const uint32_t b = (uint32_t)(__builtin_ffsll((long long)m) - 1);
    m &= m - 1ull;
    return b;
}
__device__ __forceinline__ uint64_t row_select(uint32_t row, uint64_t m0, uint64_t m1, uint64_t m2, uint64_t m3) {
    return row == 0 ? m0 : (row == 1 ? m1 : (row == 2 ? m2 : m3));
}

// Compacted row lists (round 2).  Instead of carrying 64-bit sub-list masks in VGPRs and peeling one bit per iteration
// (v_ffbl x2, min3, 64-bit and / add, selects, a ballot + branch for "any lane has more": ~22 VALU + SALU per iteration,
// half of K6's inner loop), every lane appends its staged entry's index to the lists of the rows (4x4 blocks) it
// overlaps: position = entries of that row so far + v_mbcnt of the row's ballot.  The lists (uint16 slice indices,
// pre-filled with the null entry so that a row that runs out keeps reading harmless entries) live in LDS, wave-private;
// the inner loop is a COUNTED loop to the longest row list of the wave: one ds_read_b64 of four indices per four
// iterations, one v_bfe per iteration.  Rows of a wave re-synchronise only per slice, with no refill logic.
struct alignas(16) RowLists {
    uint16_t idx[GDR_BLOCK / GDR_WAVE][4][GDR_BLOCK];   // [wave][row][position]
    uint16_t pad[8];   // (null entries; the read-ahead is clamped to the row's own list: no wave reads another wave's lists)
};

// pre-fill this wave's four lists with the null entry (2 x 16 bytes per lane = 2 KiB)
__device__ __forceinline__ void row_lists_clear(RowLists& rl, uint32_t wave) {
    uint4* p = reinterpret_cast<uint4*>(&rl.idx[wave][0][0]);
    const uint32_t nn = (uint32_t)GDR_NULL_ENTRY | ((uint32_t)GDR_NULL_ENTRY << 16);
    p[lane_id()] = make_uint4(nn, nn, nn, nn);
    p[GDR_WAVE + lane_id()] = make_uint4(nn, nn, nn, nn);
}

// append group g's entries (one per lane) to the row lists; n[r] = list lengths so far (wave-uniform)
// (mine[r] = this lane's own vote in m[r]: handed over instead of being fished out of the 64-bit mask again)
__device__ __forceinline__ void row_lists_append(RowLists& rl, uint32_t wave, int g, uint64_t m0, uint64_t m1, uint64_t m2,
                                                 uint64_t m3, const bool (&mine)[4], int (&n)[4]) {
    const uint32_t lane = lane_id();
    const uint16_t e = (uint16_t)(g * GDR_WAVE + (int)lane);
    const uint64_t m[4] = {m0, m1, m2, m3};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m[r] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[r], (uint32_t)n[r]));
        if (mine[r]) rl.idx[wave][r][pos] = e;
        n[r] += __popcll(m[r]);
    }
}

__device__ __forceinline__ void wave_lds_fence() {   // wave-private LDS data written by some lanes, read by others
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

}  // namespace

// segments of cut tile lists (see tile_order_kernel): arguments shared by the K6 / K7 launches
static inline int seg_rounds_of(const gdr_binning* bin, const gdr_image* img) {
    return (img->seg_base && bin->seg_len > 0) ? bin->seg_len / GDR_BLOCK : 0;
}
#define GDR_SEG_FWD_ARGS(bin, img) (img)->seg_base, (bin)->seg_state, seg_rounds_of(bin, img)
#define GDR_SEG_BWD_ARGS(bin, img)                                                                          \
    (img)->seg_base, (const float*)(bin)->seg_state, (const uint2*)(bin)->seg_extra, (bin)->seg_count,     \
        seg_rounds_of(bin, img), (seg_rounds_of(bin, img) ? (bin)->seg_cap : 0)
#define GDR_BWD_GRID(bin, img, ntiles) dim3((unsigned)((ntiles) + (seg_rounds_of(bin, img) ? (bin)->seg_cap : 0)))

}  // namespace gdr
