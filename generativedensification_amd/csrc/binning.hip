// binning.hip — K2..K5 of the rasterizer for gfx950 (integer / byte work, HBM-bound).
//   K2  exclusive scan of the per-block sums of tiles_touched (the per-Gaussian part of
//       the scan is recomputed inside K3 from tiles_touched, so no N-long offsets array
//       is ever written or read)
//   K3  duplicate_with_keys: key = (tile << 32) | float_bits(depth), value = Gaussian id
//   K4  stable LSD radix sort of the D (u64 key, u32 value) pairs on the low
//       32 + ceil(log2(tiles)) bits, 8 bits per pass: histogram / row scan / scatter
//   K5  identify_tile_ranges
// Behaviour: SURVEY.md Appendix A.2.  Equal keys keep emission order (ascending Gaussian
// index), exactly like the reference's CUB DeviceRadixSort, so the sorted list is
// bit-identical to oracle_bin() in oracle/gdr_oracle.c.
//
// CDNA4 notes: wave = 64, so the in-wave stable ranking uses 64-bit ballots (one per
// digit bit) instead of 32-wide match_any; LDS holds one 256-bin counter row per wave.
#include <string.h>

#include "gdr_common.h"

namespace gdr {

namespace {

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// inclusive wave scan (uint32) via cross-lane shuffles
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(v, off, 64);
        if ((int)lane_id() >= off) v += t;
    }
    return v;
}

// Block-wide exclusive scan of one value per thread (GDR_BLOCK threads). Returns the
// exclusive prefix; *total receives the block sum.  `lds` needs >= 8 uint32.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* lds, uint32_t* total) {
    const uint32_t incl = wave_incl_scan(v);
    const uint32_t w = threadIdx.x >> 6;
    if (lane_id() == 63) lds[w] = incl;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (uint32_t k = 0; k < GDR_BLOCK / GDR_WAVE; ++k)
        if (k < w) base += lds[k];
    if (total) *total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + incl - v;
}

// Duplicate count of a view.  Host-sized call: bv.D.  Device-sized call (gdr_binning.d_dev, include/gdr.h): the count
// K1 left on the device, clamped to the capacity the buffers were carved for (the caller compares the two afterwards
// and repeats the view if it did not fit), and the sort's block count derived from it.
__device__ __forceinline__ uint64_t view_D(const BinView& bv) {
    if (!bv.d_dev) return bv.D;
    const uint64_t d = *bv.d_dev;
    return d < bv.D ? d : bv.D;
}
__device__ __forceinline__ uint32_t view_nblk(const BinView& bv, uint64_t D) {
    return bv.d_dev ? (uint32_t)((D + GDR_SORT_TILE - 1) / GDR_SORT_TILE) : bv.nblk;
}

// ---------------------------------------------------------------------------------
// K2: single workgroup, exclusive scan of block_sums[0..nb) in place;
//     block_sums[nb] = num_rendered[0] = total.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(GDR_BLOCK) void scan_block_sums_kernel(uint32_t* __restrict__ block_sums,
                                                                     int nb,
                                                                     uint32_t* __restrict__ num_rendered) {
    __shared__ uint32_t lds[8];
    uint32_t carry = 0;
    for (int base = 0; base < nb; base += GDR_BLOCK) {
        const int i = base + (int)threadIdx.x;
        const uint32_t v = i < nb ? block_sums[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan(v, lds, &tot);
        if (i < nb) block_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) {
        block_sums[nb] = carry;
        *num_rendered = carry;
    }
}

// ---------------------------------------------------------------------------------
// K3
// ---------------------------------------------------------------------------------
// (every binning kernel takes the BinViews table by value and works on view blockIdx.y: the ~13 launches of the
// binning chain are issued ONCE for all views of a multi-view node — the chain is launch-rate and latency bound,
// gdr_common.h)
__global__ __launch_bounds__(GDR_BLOCK) void duplicate_kernel(const BinViews vs, int N, int gx, int tiles) {
    const BinView& bv = vs.v[blockIdx.y];
    // (ranges of empty tiles must read (0,0) after K5: cleared here, long before ranges_kernel runs — one launch less in
    // a chain whose short kernels each wait for a free slot between other views' kernels)
    for (int t = blockIdx.x * GDR_BLOCK + threadIdx.x; t < tiles; t += gridDim.x * GDR_BLOCK) bv.ranges[t] = make_uint2(0u, 0u);
    const int32_t* __restrict__ radii = bv.radii;
    const float* __restrict__ depths = bv.depths;
    const int4* __restrict__ rect = bv.rect;
    const uint32_t* __restrict__ tiles_touched = bv.tiles_touched;
    const uint32_t* __restrict__ block_offsets = bv.block_offs;
    uint64_t* __restrict__ keys = bv.keys[0];
    uint32_t* __restrict__ vals = bv.vals[0];
    const uint64_t D = bv.D;   // (the capacity of the buffers in a device-sized call)
    __shared__ uint32_t lds[8];
    const int i = blockIdx.x * GDR_BLOCK + threadIdx.x;
    const uint32_t t = i < N ? tiles_touched[i] : 0u;
    uint32_t off = block_offsets[blockIdx.x] + block_excl_scan(t, lds, nullptr);
    if (t == 0 || radii[i] <= 0) return;
    const int4 r = rect[i];
    const uint64_t dbits = (uint64_t)__float_as_uint(depths[i]);
    for (int y = r.y; y < r.w; ++y)
        for (int x = r.x; x < r.z; ++x) {
            if (off < D) {  // capacity guard: never write past the caller's buffers
                keys[off] = ((uint64_t)(uint32_t)(y * gx + x) << 32) | dbits;
                vals[off] = (uint32_t)i;
            }
            ++off;
        }
}

// ---------------------------------------------------------------------------------
// K4: one radix pass = histogram -> row scan -> scatter.
// hist layout: hist[digit * nblk + blk]; totals[digit] after it (GDR_RADIX entries).
// Key order inside a workgroup tile: (wave, round, lane) == ascending index => stable.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t tile_index(uint32_t blk, uint32_t wave, int round) {
    return (uint64_t)blk * GDR_SORT_TILE + (uint64_t)wave * (GDR_WAVE * GDR_SORT_ITEMS) +
           (uint64_t)round * GDR_WAVE + lane_id();
}

__global__ __launch_bounds__(GDR_BLOCK) void sort_hist_kernel(const BinViews vs, int cur, int shift) {
    const BinView& bv = vs.v[blockIdx.y];
    const uint64_t D = view_D(bv);
    const uint32_t nblk = view_nblk(bv, D);
    if (blockIdx.x >= nblk) return;   // (the grid is sized for the view with the most duplicates / for the capacity)
    const uint64_t* __restrict__ keys = bv.keys[cur];
    uint32_t* __restrict__ hist = bv.hist;
    __shared__ uint32_t cnt[GDR_BLOCK / GDR_WAVE][GDR_RADIX];
    for (int k = threadIdx.x; k < (GDR_BLOCK / GDR_WAVE) * GDR_RADIX; k += GDR_BLOCK)
        (&cnt[0][0])[k] = 0;
    __syncthreads();
    const uint32_t w = threadIdx.x >> 6;
#pragma unroll 4
    for (int j = 0; j < GDR_SORT_ITEMS; ++j) {
        const uint64_t idx = tile_index(blockIdx.x, w, j);
        if (idx < D) {
            const uint32_t d = (uint32_t)(keys[idx] >> shift) & (GDR_RADIX - 1);
            atomicAdd(&cnt[w][d], 1u);
        }
    }
    __syncthreads();
    const uint32_t d = threadIdx.x;  // GDR_BLOCK == GDR_RADIX
    hist[(uint64_t)d * nblk + blockIdx.x] = cnt[0][d] + cnt[1][d] + cnt[2][d] + cnt[3][d];
}

// one workgroup per digit: exclusive scan of its row of nblk per-block counts.
__global__ __launch_bounds__(GDR_BLOCK) void sort_rowscan_kernel(const BinViews vs) {
    const BinView& bv = vs.v[blockIdx.y];
    const uint32_t nblk = view_nblk(bv, view_D(bv));
    uint32_t* __restrict__ hist = bv.hist;
    uint32_t* __restrict__ totals = bv.hist + (uint64_t)nblk * GDR_RADIX;
    if (nblk == 0) return;
    __shared__ uint32_t lds[8];
    uint32_t* row = hist + (uint64_t)blockIdx.x * nblk;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nblk; base += GDR_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nblk ? row[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan(v, lds, &tot);
        if (i < nblk) row[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// STAGED (the first pass of the tile partition, whose digits — the low 8 bits of the tile id — are spread evenly: ~16 keys
// per digit and workgroup): the workgroup's keys are first put in digit order in LDS and then written out by consecutive
// threads, so that a wave writes a few runs of consecutive pairs instead of 64 scattered 8- and 4-byte words.
template <bool STAGED>
__global__ __launch_bounds__(GDR_BLOCK) void sort_scatter_kernel(const BinViews vs, int cur, int shift) {
    const BinView& bv = vs.v[blockIdx.y];
    const uint64_t D = view_D(bv);
    const uint32_t nblk = view_nblk(bv, D);
    if (blockIdx.x >= nblk) return;
    const uint64_t* __restrict__ keys_in = bv.keys[cur];
    const uint32_t* __restrict__ vals_in = bv.vals[cur];
    uint64_t* __restrict__ keys_out = bv.keys[cur ^ 1];
    uint32_t* __restrict__ vals_out = bv.vals[cur ^ 1];
    const uint32_t* __restrict__ hist = bv.hist;
    const uint32_t* __restrict__ totals = bv.hist + (uint64_t)nblk * GDR_RADIX;
    __shared__ uint32_t cnt[GDR_BLOCK / GDR_WAVE][GDR_RADIX];
    __shared__ uint32_t lds[8];
    for (int k = threadIdx.x; k < (GDR_BLOCK / GDR_WAVE) * GDR_RADIX; k += GDR_BLOCK)
        (&cnt[0][0])[k] = 0;
    // global base of digit d = exclusive scan of the digit totals
    const uint32_t digit_base = block_excl_scan(totals[threadIdx.x], lds, nullptr);  // includes a barrier

    const uint32_t w = threadIdx.x >> 6;
    uint64_t key[GDR_SORT_ITEMS];
    uint32_t val[GDR_SORT_ITEMS];
    uint32_t rank[GDR_SORT_ITEMS];
#pragma unroll
    for (int j = 0; j < GDR_SORT_ITEMS; ++j) {
        const uint64_t idx = tile_index(blockIdx.x, w, j);
        const bool valid = idx < D;
        key[j] = valid ? keys_in[idx] : ~0ull;
        val[j] = valid ? vals_in[idx] : 0u;
    }
#pragma unroll
    for (int j = 0; j < GDR_SORT_ITEMS; ++j) {
        const uint64_t idx = tile_index(blockIdx.x, w, j);
        const bool valid = idx < D;
        const uint32_t d = (uint32_t)(key[j] >> shift) & (GDR_RADIX - 1);
        // lanes of this wave holding the same digit (64-wide match via one ballot per bit)
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < GDR_RADIX_BITS; ++b) {
            const uint64_t bal = __ballot((d >> b) & 1u);
            m &= ((d >> b) & 1u) ? bal : ~bal;
        }
        uint32_t prior = 0;
        if (valid) prior = cnt[w][d];
        const uint32_t below = (uint32_t)__popcll(m & lanemask_lt());
        rank[j] = prior + below;
        // the highest lane of each group publishes the new count (all reads above are done:
        // a wave's DS operations execute in order)
        if (valid && (m >> lane_id()) == 1ull) cnt[w][d] = prior + below + 1u;
    }
    __syncthreads();
    if constexpr (STAGED) {
        __shared__ uint64_t s_key[GDR_SORT_TILE];
        __shared__ uint32_t s_val[GDR_SORT_TILE];
        __shared__ uint32_t s_delta[GDR_RADIX];   // global position - position in the workgroup's digit-ordered list
        {
            const uint32_t d = threadIdx.x;
            uint32_t tot = 0;
#pragma unroll
            for (int k = 0; k < GDR_BLOCK / GDR_WAVE; ++k) tot += cnt[k][d];
            uint32_t lbase = block_excl_scan(tot, lds, nullptr);   // where digit d starts in the local list
            s_delta[d] = digit_base + hist[(uint64_t)d * nblk + blockIdx.x] - lbase;
#pragma unroll
            for (int k = 0; k < GDR_BLOCK / GDR_WAVE; ++k) {
                const uint32_t c = cnt[k][d];
                cnt[k][d] = lbase;
                lbase += c;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < GDR_SORT_ITEMS; ++j) {
            const uint64_t idx = tile_index(blockIdx.x, w, j);
            if (idx < D) {
                const uint32_t d = (uint32_t)(key[j] >> shift) & (GDR_RADIX - 1);
                const uint32_t lpos = cnt[w][d] + rank[j];
                s_key[lpos] = key[j];
                s_val[lpos] = val[j];
            }
        }
        __syncthreads();
        const uint64_t first = (uint64_t)blockIdx.x * GDR_SORT_TILE;
        const uint32_t n = (uint32_t)(D - first < (uint64_t)GDR_SORT_TILE ? D - first : (uint64_t)GDR_SORT_TILE);
        for (uint32_t i = threadIdx.x; i < n; i += GDR_BLOCK) {
            const uint64_t k = s_key[i];
            const uint32_t pos = i + s_delta[(uint32_t)(k >> shift) & (GDR_RADIX - 1)];
            keys_out[pos] = k;
            vals_out[pos] = s_val[i];
        }
        return;
    }
    {   // per-digit: turn per-wave counts into global start positions
        const uint32_t d = threadIdx.x;
        uint32_t base = digit_base + hist[(uint64_t)d * nblk + blockIdx.x];
#pragma unroll
        for (int k = 0; k < GDR_BLOCK / GDR_WAVE; ++k) {
            const uint32_t c = cnt[k][d];
            cnt[k][d] = base;
            base += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < GDR_SORT_ITEMS; ++j) {
        const uint64_t idx = tile_index(blockIdx.x, w, j);
        if (idx < D) {
            const uint32_t d = (uint32_t)(key[j] >> shift) & (GDR_RADIX - 1);
            const uint32_t pos = cnt[w][d] + rank[j];
            keys_out[pos] = key[j];
            vals_out[pos] = val[j];
        }
    }
}

// ---------------------------------------------------------------------------------
// K5
// ---------------------------------------------------------------------------------
// (ranges of empty tiles must read (0,0): cleared by duplicate_kernel; by this kernel when there is nothing to duplicate)
__global__ __launch_bounds__(GDR_BLOCK) void ranges_clear_kernel(const BinViews vs, int tiles) {
    const int t = blockIdx.x * GDR_BLOCK + threadIdx.x;
    if (t < tiles) vs.v[blockIdx.y].ranges[t] = make_uint2(0u, 0u);
}

__global__ __launch_bounds__(GDR_BLOCK) void ranges_kernel(const BinViews vs, int cur) {
    const BinView& bv = vs.v[blockIdx.y];
    const uint64_t* __restrict__ keys = bv.keys[cur];
    const uint64_t D = view_D(bv);
    uint2* __restrict__ ranges = bv.ranges;
    const uint64_t e = (uint64_t)blockIdx.x * GDR_BLOCK + threadIdx.x;
    if (e >= D) return;
    const uint32_t t = (uint32_t)(keys[e] >> 32);
    if (e == 0) {
        ranges[t].x = 0;
    } else {
        const uint32_t tp = (uint32_t)(keys[e - 1] >> 32);
        if (tp != t) {
            ranges[tp].y = (uint32_t)e;
            ranges[t].x = (uint32_t)e;
        }
    }
    if (e == D - 1) ranges[t].y = (uint32_t)D;
}


// =================================================================================
// Direct tile binning (round 3, the default for images of <= GDR_BIN_MAX_TILES tiles).  The list only has to end up
// PARTITIONED by tile — the per-tile depth sort that follows orders every list by (depth, id) whatever order it
// arrives in — so the partition is a counting sort on the tile id done straight from the Gaussians' tile rects, with
// no (key, value) stream written and re-read:
//   tile_count    every workgroup takes a fixed chunk of Gaussians, counts the tiles their rects cover in an LDS
//                 histogram (ds_add_u32: 4 cycles, integer LDS atomics are the fast kind) and writes it as one
//                 column of the (tiles x workgroups) count matrix;
//   tile_scan     exclusive scan over the workgroups in place + the tiles' totals (tile_order_kernel, the one-workgroup
//                 kernel that follows anyway, scans the totals into the ranges: empty tiles read (0,0), as the reference's);
//   tile_scatter  the same chunks again: LDS cursor per tile = range start + this workgroup's prefix; every rect
//                 cell draws a position with ds_add_rtn and writes ONE 8-byte word (id << 32 | depth bits).
// 3 launches and ~52 bytes per Gaussian + 8 per entry instead of duplicate + 2 x (hist, row scan, scatter) + ranges =
// 8 launches and 20 + 12 + 2 x 32 + 8 bytes per entry.  Capacity-guarded like the old path (device-sized calls).
// =================================================================================
#define GDR_BIN_THREADS 1024

// The tile rects of GDR_BIN_BATCH Gaussians of a thread (i0 + j * 1024), an empty rect for a Gaussian that is not binned
// (culled, or beyond hi).  All loads are issued before the first is used: with "tiles_touched, then radii, then rect" per
// Gaussian a thread paid three dependent trips to memory per iteration and tile_count was latency, not bandwidth.
#define GDR_BIN_BATCH 4
__device__ __forceinline__ void bin_rects(const BinView& bv, int i0, int hi, int4 (&r)[GDR_BIN_BATCH]) {
    uint32_t tt[GDR_BIN_BATCH];
    int32_t rad[GDR_BIN_BATCH];
#pragma unroll
    for (int j = 0; j < GDR_BIN_BATCH; ++j) {
        const int i = i0 + j * GDR_BIN_THREADS;
        const bool in = i < hi;
        tt[j] = in ? bv.tiles_touched[i] : 0u;
        rad[j] = in ? bv.radii[i] : 0;
        r[j] = in ? bv.rect[i] : make_int4(0, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < GDR_BIN_BATCH; ++j)
        if (tt[j] == 0u || rad[j] <= 0) r[j] = make_int4(0, 0, 0, 0);
}

// count matrix layout: row w = workgroup w of tile_count / tile_scatter, `tstride` words (tiles rounded up to 64): every
// access below is coalesced — the workgroups write / read their own row, the scan walks the rows with one thread per tile
__global__ __launch_bounds__(GDR_BIN_THREADS) void tile_count_kernel(const BinViews vs, int N, int gx, int tiles, int chunk,
                                                                      int tstride) {
    const BinView& bv = vs.v[blockIdx.y];
    extern __shared__ uint32_t cnt[];
    for (int t = threadIdx.x; t < tiles; t += GDR_BIN_THREADS) cnt[t] = 0u;
    __syncthreads();
    const int lo = blockIdx.x * chunk, hi = min(N, lo + chunk);
    for (int i0 = lo + (int)threadIdx.x; i0 < hi; i0 += GDR_BIN_BATCH * GDR_BIN_THREADS) {
        int4 r[GDR_BIN_BATCH];
        bin_rects(bv, i0, hi, r);
#pragma unroll
        for (int j = 0; j < GDR_BIN_BATCH; ++j)
            for (int y = r[j].y; y < r[j].w; ++y)
                for (int x = r[j].x; x < r[j].z; ++x) atomicAdd(&cnt[y * gx + x], 1u);
    }
    __syncthreads();
    uint32_t* __restrict__ row = bv.tile_hist + (size_t)blockIdx.x * tstride;
    for (int t = threadIdx.x; t < tiles; t += GDR_BIN_THREADS) row[t] = cnt[t];
}

// Exclusive scan down the columns (over the workgroups) in place + column totals.  A workgroup takes 64 tiles x 16
// segments of the workgroup axis: thread (segment, tile) sums its segment, the segment sums are exchanged in LDS, a second
// sweep writes the prefixes.  totals: (tiles) words behind the matrix; tile_order_kernel turns them into the ranges.
#define GDR_BIN_SEGS 16
__global__ __launch_bounds__(64 * GDR_BIN_SEGS) void tile_scan_kernel(const BinViews vs, int tiles, int nwg, int tstride) {
    const BinView& bv = vs.v[blockIdx.y];
    __shared__ uint32_t segsum[GDR_BIN_SEGS][64];
    const int tl = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + tl;
    const int per = (nwg + GDR_BIN_SEGS - 1) / GDR_BIN_SEGS;
    const int w0 = min(nwg, seg * per), w1 = min(nwg, w0 + per);
    uint32_t* __restrict__ col = bv.tile_hist + t;
    uint32_t s = 0u;
    if (t < tiles)
        for (int w = w0; w < w1; ++w) s += col[(size_t)w * tstride];
    segsum[seg][tl] = s;
    __syncthreads();
    uint32_t run = 0u, total = 0u;
#pragma unroll
    for (int k = 0; k < GDR_BIN_SEGS; ++k) {
        const uint32_t v = segsum[k][tl];
        run += k < seg ? v : 0u;
        total += v;
    }
    if (t >= tiles) return;
    for (int w = w0; w < w1; ++w) {
        const uint32_t v = col[(size_t)w * tstride];
        col[(size_t)w * tstride] = run;
        run += v;
    }
    if (seg == 0) bv.tile_hist[(size_t)bv.hist_width * tstride + t] = total;
}

__global__ __launch_bounds__(GDR_BIN_THREADS) void tile_scatter_kernel(const BinViews vs, int N, int gx, int tiles, int chunk,
                                                                        int tstride) {
    const BinView& bv = vs.v[blockIdx.y];
    extern __shared__ uint32_t cur[];
    const uint2* __restrict__ ranges = bv.ranges;
    const uint32_t* __restrict__ row = bv.tile_hist + (size_t)blockIdx.x * tstride;
    for (int t = threadIdx.x; t < tiles; t += GDR_BIN_THREADS) cur[t] = ranges[t].x + row[t];
    __syncthreads();
    uint64_t* __restrict__ out = bv.keys[0];
    const uint64_t cap = bv.D;
    const int lo = blockIdx.x * chunk, hi = min(N, lo + chunk);
    for (int i0 = lo + (int)threadIdx.x; i0 < hi; i0 += GDR_BIN_BATCH * GDR_BIN_THREADS) {
        int4 r[GDR_BIN_BATCH];
        uint32_t dbits[GDR_BIN_BATCH];
#pragma unroll
        for (int j = 0; j < GDR_BIN_BATCH; ++j) {
            const int i = i0 + j * GDR_BIN_THREADS;
            dbits[j] = i < hi ? __float_as_uint(bv.depths[i]) : 0u;
        }
        bin_rects(bv, i0, hi, r);
#pragma unroll
        for (int j = 0; j < GDR_BIN_BATCH; ++j) {
            const uint64_t word = ((uint64_t)(uint32_t)(i0 + j * GDR_BIN_THREADS) << 32) | (uint64_t)dbits[j];
            for (int y = r[j].y; y < r[j].w; ++y)
                for (int x = r[j].x; x < r[j].z; ++x) {
                    const uint32_t pos = atomicAdd(&cur[y * gx + x], 1u);
                    if (pos < cap) out[pos] = word;   // capacity guard: never write past the caller's buffers
                }
        }
    }
}

// =================================================================================
// Tile-binned sort (default path).  The sort key is (tile, depth).  Instead of 6 global radix
// passes over the 12-byte pairs, the pairs are first partitioned by TILE with the stable
// passes above run on the tile bits only (2 passes at 800x800), K5 reads the tile segments off
// the partitioned keys, and then ONE workgroup per tile sorts its segment by depth in LDS with
// a stable 8-bit LSD radix sort on (depth - min depth of the tile): typically 3 passes, no
// global traffic besides one read and one write of the segment.  Entries with identical depth
// bits are finally put in ascending Gaussian-id order, which is exactly what the reference's
// single global stable sort of Gaussian-ordered duplicates yields => bit-identical sorted list.  A segment longer than the LDS capacity is sorted
// by its workgroup with the same code on the global ping-pong buffers.
// =================================================================================
// three size classes so that LDS footprint (= workgroups per CU) follows the list length:
//   short  (L <= 2048): 20 KiB, 4 waves, one workgroup per tile;
//   medium (L <= 4096): 40 KiB, 8 waves   } small grids that walk the tiles longest-first
//   long   (L <= 16384 in LDS, beyond that on the global ping-pong buffers): 144 KiB, 16 waves, 16 elements per lane }
//          (8192 / 80 KiB until object-like scenes were measured: their 10-19 k-entry lists took the global route; C4 shell
//          963 -> 991, C3 shell 3120 -> 3210, uniform scenes unchanged)
// (lists are sorted IN PLACE in LDS, tile_sort_pass_lds: 8 bytes per entry)
// (GDR_TSORT_SMALL / MEDIUM / LARGE = 2048 / 4096 / 16384: gdr_common.h)

struct TileSortBufs {
    uint32_t *kA, *vA, *kB, *vB;
};

// one stable 8-bit pass over [0,L) from (kA,vA) to (kB,vB); cnt: [NW][256] LDS counters (NW waves).
// bstart (optional, 257 entries): receives the exclusive scan of the digit totals (= where each digit's run
// starts in the output) and L at [256].
// The source is an accessor, kv_at(i) -> (key, value): two arrays, or the packed 64-bit words of the direct tile binning.
// These passes run on lists that do not fit in LDS — one workgroup, a few dozen iterations per lane, every one a trip to
// HBM: B elements per lane are loaded before the first is used (C4 `shell`, 22 lists of 16-20 k entries: the long-class
// launch 175 -> see DESIGN.md §3).
template <int NW, int B, class KV>
__device__ __forceinline__ void tile_sort_pass_kv(KV kv_at, uint32_t* kB, uint32_t* vB, uint32_t L,
                                                  uint32_t kmin, int shift, uint32_t (*cnt)[GDR_RADIX], uint32_t* scan_lds,
                                                  uint32_t* bstart = nullptr) {
    const uint32_t w = threadIdx.x >> 6, lane = lane_id();
    const uint32_t Lw = ((L + NW * GDR_WAVE - 1) / (NW * GDR_WAVE)) * GDR_WAVE;
    const uint32_t c0 = min(L, w * Lw), c1 = min(L, c0 + Lw);
    for (int k = threadIdx.x; k < NW * GDR_RADIX; k += NW * GDR_WAVE) (&cnt[0][0])[k] = 0;
    __syncthreads();
    for (uint32_t i0 = c0; i0 < c1; i0 += GDR_WAVE * B) {
        uint32_t k[B];
#pragma unroll
        for (int j = 0; j < B; ++j) {
            const uint32_t i = i0 + (uint32_t)j * GDR_WAVE + lane;
            k[j] = i < c1 ? kv_at(i).x : 0u;
        }
#pragma unroll
        for (int j = 0; j < B; ++j)
            if (i0 + (uint32_t)j * GDR_WAVE + lane < c1) atomicAdd(&cnt[w][((k[j] - kmin) >> shift) & (GDR_RADIX - 1)], 1u);
    }
    __syncthreads();
    {   // the first 256 threads own one digit each: per-wave offsets + exclusive scan of the digit totals
        const uint32_t d = threadIdx.x & (GDR_RADIX - 1);
        const bool owner = threadIdx.x < GDR_RADIX;
        uint32_t tot = 0;
        if (owner)
            for (int k = 0; k < NW; ++k) tot += cnt[k][d];
        const uint32_t incl = wave_incl_scan(owner ? tot : 0u);
        if (owner && lane == 63) scan_lds[w] = incl;
        __syncthreads();
        if (owner) {
            uint32_t base = incl - tot;
            for (uint32_t k = 0; k < w; ++k) base += scan_lds[k];
            if (bstart) {
                bstart[d] = base;
                if (d == 0) bstart[GDR_RADIX] = L;
            }
            for (int k = 0; k < NW; ++k) { const uint32_t c = cnt[k][d]; cnt[k][d] = base; base += c; }
        }
    }
    __syncthreads();
    for (uint32_t i0 = c0; i0 < c1; i0 += GDR_WAVE * B) {
        uint2 e[B];
#pragma unroll
        for (int j = 0; j < B; ++j) {
            const uint32_t i = i0 + (uint32_t)j * GDR_WAVE + lane;
            e[j] = i < c1 ? kv_at(i) : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int j = 0; j < B; ++j) {
            if (i0 + (uint32_t)j * GDR_WAVE >= c1) break;   // uniform over the wave
            const bool valid = i0 + (uint32_t)j * GDR_WAVE + lane < c1;
            const uint32_t d = ((e[j].x - kmin) >> shift) & (GDR_RADIX - 1);
            uint64_t m = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < GDR_RADIX_BITS; ++bit) {
                const uint64_t bal = __ballot((d >> bit) & 1u);
                m &= ((d >> bit) & 1u) ? bal : ~bal;
            }
            uint32_t prior = 0;
            if (valid) prior = cnt[w][d];
            const uint32_t below = (uint32_t)__popcll(m & lanemask_lt());
            if (valid) {
                kB[prior + below] = e[j].x;
                vB[prior + below] = e[j].y;
                if ((m >> lane) == 1ull) cnt[w][d] = prior + below + 1u;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();  // (also the workgroup-scope release/acquire of the global stores)
}

template <int NW>
__device__ __forceinline__ void tile_sort_pass(const TileSortBufs& b, uint32_t L, uint32_t kmin, int shift,
                                               uint32_t (*cnt)[GDR_RADIX], uint32_t* scan_lds,
                                               uint32_t* bstart = nullptr) {
    const uint32_t* kA = b.kA;
    const uint32_t* vA = b.vA;
    tile_sort_pass_kv<NW, 4>([kA, vA](uint32_t i) { return make_uint2(kA[i], vA[i]); }, b.kB, b.vB, L, kmin, shift, cnt,
                             scan_lds, bstart);
}

// The same stable 8-bit pass IN PLACE on an LDS-resident list (kA, vA; kA == vA allowed: ids sorted by themselves):
// every lane first takes the <= EPT elements of its wave's chunk it is responsible for into registers, so that after
// the counting barrier nobody reads the list any more and the ranked elements can be written straight back — no second
// buffer: 8 bytes of LDS per entry instead of 16, twice the workgroups per CU (the binning chains of a multi-view node run
// next to other views' K6, which leaves ~50 KB of LDS per CU).  L <= NW * 64 * EPT.
template <int NW, int EPT>
__device__ __forceinline__ void tile_sort_pass_lds(uint32_t* kA, uint32_t* vA, uint32_t L, uint32_t kmin, int shift,
                                                   uint32_t (*cnt)[GDR_RADIX], uint32_t* scan_lds) {
    const uint32_t w = threadIdx.x >> 6, lane = lane_id();
    const uint32_t Lw = ((L + NW * GDR_WAVE - 1) / (NW * GDR_WAVE)) * GDR_WAVE;
    const uint32_t c0 = min(L, w * Lw), c1 = min(L, c0 + Lw);
    for (int k = threadIdx.x; k < NW * GDR_RADIX; k += NW * GDR_WAVE) (&cnt[0][0])[k] = 0;
    __syncthreads();
    uint32_t key[EPT], val[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const uint32_t i = c0 + (uint32_t)j * GDR_WAVE + lane;
        const bool valid = i < c1;
        key[j] = valid ? kA[i] : 0u;
        val[j] = valid ? vA[i] : 0u;
        if (valid) atomicAdd(&cnt[w][((key[j] - kmin) >> shift) & (GDR_RADIX - 1)], 1u);
    }
    __syncthreads();
    {   // the first 256 threads own one digit each: per-wave offsets + exclusive scan of the digit totals
        const uint32_t d = threadIdx.x & (GDR_RADIX - 1);
        const bool owner = threadIdx.x < GDR_RADIX;
        uint32_t tot = 0;
        if (owner)
            for (int k = 0; k < NW; ++k) tot += cnt[k][d];
        const uint32_t incl = wave_incl_scan(owner ? tot : 0u);
        if (owner && lane == 63) scan_lds[w] = incl;
        __syncthreads();
        if (owner) {
            uint32_t base = incl - tot;
            for (uint32_t k = 0; k < w; ++k) base += scan_lds[k];
            for (int k = 0; k < NW; ++k) { const uint32_t c = cnt[k][d]; cnt[k][d] = base; base += c; }
        }
    }
    __syncthreads();   // (every element of the list is in a register by now: the writes below overwrite nothing unread)
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const uint32_t i0 = c0 + (uint32_t)j * GDR_WAVE;
        if (i0 >= c1) break;   // uniform over the wave
        const bool valid = i0 + lane < c1;
        const uint32_t d = ((key[j] - kmin) >> shift) & (GDR_RADIX - 1);
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < GDR_RADIX_BITS; ++bit) {
            const uint64_t bal = __ballot((d >> bit) & 1u);
            m &= ((d >> bit) & 1u) ? bal : ~bal;
        }
        uint32_t prior = 0;
        if (valid) prior = cnt[w][d];
        const uint32_t below = (uint32_t)__popcll(m & lanemask_lt());
        if (valid) {
            kA[prior + below] = key[j];
            vA[prior + below] = val[j];
            if ((m >> lane) == 1ull) cnt[w][d] = prior + below + 1u;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
}

// ties on identical depth bits: ascending Gaussian id — the order the reference's stable sort of
// (Gaussian-ordered) emission leaves them in; our emission order is arbitrary (block_offs).
// Runs of up to GDR_TIE_SERIAL equal keys are insertion-sorted by the thread that finds their head; longer runs
// (thousands of Gaussians at one depth: a grid plane seen head-on) are queued and then sorted by the whole
// workgroup with four stable 8-bit passes on the ids, vA -> vB -> vA -> vB -> vA.  runs: 2*GDR_TIE_RUNS+1 LDS words.
#define GDR_TIE_SERIAL 96
#define GDR_TIE_RUNS 96
// EPT > 0: the list is LDS-resident, long runs are sorted in place (tile_sort_pass_lds), vB is not used.
template <int NW, int EPT = 0>
__device__ __forceinline__ void tile_sort_ties(const uint32_t* kA, uint32_t* vA, uint32_t* vB, uint32_t L,
                                               uint32_t (*cnt)[GDR_RADIX], uint32_t* scan_lds, uint32_t* runs) {
    constexpr uint32_t NT = NW * GDR_WAVE;
    if (threadIdx.x == 0) runs[2 * GDR_TIE_RUNS] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i + 1 < L; i += NT) {
        if (kA[i] == kA[i + 1] && (i == 0 || kA[i - 1] != kA[i])) {
            uint32_t j = i + 1;
            while (j < L && kA[j] == kA[i]) ++j;
            if (j - i > GDR_TIE_SERIAL) {
                const uint32_t slot = atomicAdd(&runs[2 * GDR_TIE_RUNS], 1u);
                if (slot < GDR_TIE_RUNS) { runs[2 * slot] = i; runs[2 * slot + 1] = j - i; continue; }
            }
            for (uint32_t a = i + 1; a < j; ++a) {  // insertion sort of the run [i, j) by id
                const uint32_t x = vA[a];
                uint32_t c = a;
                while (c > i && vA[c - 1] > x) { vA[c] = vA[c - 1]; --c; }
                vA[c] = x;
            }
        }
    }
    __syncthreads();
    const uint32_t nruns = min(runs[2 * GDR_TIE_RUNS], (uint32_t)GDR_TIE_RUNS);
    for (uint32_t r = 0; r < nruns; ++r) {  // uniform over the workgroup
        const uint32_t i0 = runs[2 * r], Lr = runs[2 * r + 1];
        if constexpr (EPT > 0) {
            for (int shift = 0; shift < 32; shift += GDR_RADIX_BITS)
                tile_sort_pass_lds<NW, EPT>(vA + i0, vA + i0, Lr, 0u, shift, cnt, scan_lds);
            continue;
        }
        TileSortBufs t;
        t.kA = t.vA = vA + i0;
        t.kB = t.vB = vB + i0;
        for (int shift = 0; shift < 32; shift += GDR_RADIX_BITS) {
            tile_sort_pass<NW>(t, Lr, 0u, shift, cnt, scan_lds);
            uint32_t* x = t.kA; t.kA = t.vA = t.kB; t.kB = t.vB = x;
        }
    }
}

// keys_part: tile-partitioned u64 keys (tile << 32 | depth); vals_part: ids; outputs sorted.
// scratch32: 2*D uint32 of global scratch (depth keys of the tiles that do not fit in LDS)
// handles tiles with LMIN < L <= CAP in LDS; NW waves per workgroup.  TOP also takes the lists longer than CAP:
// one stable pass on the TOP 8 bits of (depth - min depth) through global memory splits the list into 256
// ordered buckets, then runs of whole buckets that fit are sorted in LDS like short lists (a single bucket
// larger than CAP — thousands of near-identical depths in one tile — is finished by LSD passes on the global
// buffers).  3 reads + 2 writes of the list instead of one global round trip per 8 key bits.
// PACKED: the partitioned list is one word per entry, (id << 32) | depth bits (direct tile binning); otherwise keys
// (tile << 32 | depth bits) + values in two arrays (radix partition).  Output: the sorted ids only — the sorted keys are
// a function of (ranges, ids, depths) and nothing downstream reads them (the parity tests rebuild them from those).
template <int CAP, int LMIN, int NW, bool TOP, bool PACKED>
__global__ __launch_bounds__(NW * GDR_WAVE) void tile_sort_kernel(const BinViews vs, int in, int ntiles) {
    const BinView& bv = vs.v[blockIdx.y];
    const uint2* __restrict__ ranges = bv.ranges;
    const uint64_t* __restrict__ keys_part = bv.keys[in];
    uint32_t* __restrict__ vals_part = bv.vals[in];
    uint32_t* __restrict__ vals_out = bv.vals[in ^ 1];
    uint32_t* __restrict__ scratch32 = bv.scratch32;
    const uint64_t D = bv.D;   // offset of the second scratch half: the carved size, not the live count
    const uint32_t* __restrict__ tile_order = bv.tile_order;
    if (view_D(bv) == 0) return;
    constexpr uint32_t NT = NW * GDR_WAVE;
    __shared__ uint32_t lds_elems[2 * CAP];   // (depth key, id) of a list of <= CAP entries, sorted in place
    __shared__ uint32_t cnt[NW][GDR_RADIX];
    __shared__ uint32_t misc[8];
    __shared__ uint32_t mm[2 * NW];
    __shared__ uint32_t bstart[TOP ? GDR_RADIX + 1 : 1];
    __shared__ uint32_t bstart2[TOP ? GDR_RADIX + 1 : 1];
    __shared__ uint32_t tie_runs[2 * GDR_TIE_RUNS + 1];
    const uint32_t w = threadIdx.x >> 6, lane = lane_id();

    // workgroup reduction of (min, max) over the values each thread collected
    auto min_max = [&](uint32_t& kmin, uint32_t& kmax) __attribute__((always_inline)) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, 64));
            kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, 64));
        }
        if (lane == 0) { mm[w] = kmin; mm[NW + w] = kmax; }
        __syncthreads();
        for (int k = 0; k < NW; ++k) { kmin = min(kmin, mm[k]); kmax = max(kmax, mm[NW + k]); }
    };
    // Lc <= CAP pairs (key_at(i), vsrc[i]) -> sorted at keys_out / vals_out [o, o + Lc)
    auto sort_in_lds = [&](auto key_at, auto val_at, uint32_t o, uint32_t Lc) __attribute__((always_inline)) {
        __syncthreads();  // LDS reuse
        constexpr int EPT = CAP / (int)NT;
        uint32_t* const kA = lds_elems;
        uint32_t* const vA = lds_elems + CAP;
        uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
        for (uint32_t i = threadIdx.x; i < Lc; i += NT) {
            const uint32_t k = key_at(i);
            kA[i] = k;
            vA[i] = val_at(i);
            kmin = min(kmin, k);
            kmax = max(kmax, k);
        }
        min_max(kmin, kmax);
        const uint32_t span = kmax - kmin;
        const int nbits = span ? 32 - __builtin_clz(span) : 0;
        for (int shift = 0; shift < nbits; shift += GDR_RADIX_BITS)
            tile_sort_pass_lds<NW, EPT>(kA, vA, Lc, kmin, shift, cnt, misc);
        tile_sort_ties<NW, EPT>(kA, vA, nullptr, Lc, cnt, misc, tie_runs);
        for (uint32_t i = threadIdx.x; i < Lc; i += NT) vals_out[o + i] = vA[i];
    };

    // LMIN > 0 (long-list class): a small grid walks the tiles longest-first (tile_order is sorted by
    // list length / 16, descending) and stops at the first tile that is clearly in the other class
    for (uint32_t slot = blockIdx.x; slot < (uint32_t)ntiles; slot += gridDim.x) {
        const uint32_t tile = LMIN > 0 ? tile_order[slot] : slot;
        const uint2 rg = ranges[tile];
        const uint32_t L = rg.y - rg.x;
        if (LMIN > 0 && (L >> 4) < ((uint32_t)LMIN >> 4)) return;          // every later tile is shorter
        if (L <= (uint32_t)LMIN || (!TOP && L > (uint32_t)CAP)) continue;  // other size class
        if (L <= (uint32_t)CAP) {
            const uint64_t* kp = keys_part + rg.x;
            const uint32_t* vp = vals_part + rg.x;
            if constexpr (PACKED)
                sort_in_lds([kp](uint32_t i) { return (uint32_t)kp[i]; }, [kp](uint32_t i) { return (uint32_t)(kp[i] >> 32); }, rg.x, L);
            else
                sort_in_lds([kp](uint32_t i) { return (uint32_t)kp[i]; }, [vp](uint32_t i) { return vp[i]; }, rg.x, L);
            continue;
        }
        if constexpr (TOP) {
            // this tile's own slices of the scratch / value buffers
            uint32_t* const kA = scratch32 + rg.x;
            uint32_t* const kB = scratch32 + D + rg.x;
            uint32_t* const vA = vals_part + rg.x;
            uint32_t* const vB = vals_out + rg.x;
            __syncthreads();  // mm / bstart reuse
            uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
            for (uint32_t i = threadIdx.x; i < L; i += NT) {
                const uint32_t k = (uint32_t)keys_part[rg.x + i];
                if constexpr (!PACKED) kA[i] = k;   // (packed: the bucket pass below reads the packed words themselves)
                kmin = min(kmin, k);
                kmax = max(kmax, k);
            }
            min_max(kmin, kmax);
            const uint32_t span = kmax - kmin;
            const int nbits = span ? 32 - __builtin_clz(span) : 0;
            // last resort for a bucket that is still longer than CAP: LSD passes on its low `low_bits` bits on the
            // global buffers (g.kA/g.vA hold it, g.kB/g.vB are its scratch), ties, store at [o, o + Lc)
            auto finish_global = [&](TileSortBufs g, uint32_t Lc, int low_bits, uint32_t o) __attribute__((always_inline)) {
                __syncthreads();
                for (int shift = 0; shift < low_bits; shift += GDR_RADIX_BITS) {
                    tile_sort_pass<NW>(g, Lc, kmin, shift, cnt, misc);
                    uint32_t* t = g.kA; g.kA = g.kB; g.kB = t;
                    t = g.vA; g.vA = g.vB; g.vB = t;
                }
                tile_sort_ties<NW>(g.kA, g.vA, g.vB, Lc, cnt, misc, tie_runs);
                if (g.vA != vals_out + o)
                    for (uint32_t i = threadIdx.x; i < Lc; i += NT) vals_out[o + i] = g.vA[i];
            };
            const int sh1 = nbits > GDR_RADIX_BITS ? nbits - GDR_RADIX_BITS : 0;
            if constexpr (PACKED) {   // (kB, vB): 256 ordered buckets
                const uint64_t* kp = keys_part + rg.x;
                tile_sort_pass_kv<NW, 4>([kp](uint32_t i) { const uint64_t w = kp[i]; return make_uint2((uint32_t)w, (uint32_t)(w >> 32)); },
                                         kB, vB, L, kmin, sh1, cnt, misc, bstart);
            } else {
                TileSortBufs g;
                g.kA = kA; g.vA = vA; g.kB = kB; g.vB = vB;
                tile_sort_pass<NW>(g, L, kmin, sh1, cnt, misc, bstart);
            }
            uint32_t d0 = 0;
            while (d0 < GDR_RADIX) {  // uniform over the workgroup: bstart is read-only from here on
                const uint32_t s0 = bstart[d0];
                uint32_t d1 = d0 + 1;
                while (d1 < GDR_RADIX && bstart[d1 + 1] - s0 <= (uint32_t)CAP) ++d1;
                const uint32_t Lc = bstart[d1] - s0;
                d0 = d1;
                if (Lc == 0) continue;
                if (Lc <= (uint32_t)CAP) {
                    const uint32_t* kc = kB + s0;
                    const uint32_t* vc = vB + s0;
                    sort_in_lds([kc](uint32_t i) { return kc[i]; }, [vc](uint32_t i) { return vc[i]; }, rg.x + s0, Lc);
                    continue;
                }
                TileSortBufs g;  // one bucket longer than CAP
                g.kA = kB + s0; g.vA = vB + s0; g.kB = kA + s0; g.vB = vA + s0;
                if (sh1 == 0) { finish_global(g, Lc, 0, rg.x + s0); continue; }  // one depth value: ties only
                // second level: the next 8 bits of this bucket, (kB, vB) -> (kA, vA)
                const int sh2 = sh1 > GDR_RADIX_BITS ? sh1 - GDR_RADIX_BITS : 0;
                __syncthreads();
                tile_sort_pass<NW>(g, Lc, kmin, sh2, cnt, misc, bstart2);
                uint32_t e0 = 0;
                while (e0 < GDR_RADIX) {
                    const uint32_t s1 = bstart2[e0];
                    uint32_t e1 = e0 + 1;
                    while (e1 < GDR_RADIX && bstart2[e1 + 1] - s1 <= (uint32_t)CAP) ++e1;
                    const uint32_t Lcc = bstart2[e1] - s1;
                    e0 = e1;
                    if (Lcc == 0) continue;
                    if (Lcc <= (uint32_t)CAP) {
                        const uint32_t* kc = kA + s0 + s1;
                        const uint32_t* vc = vA + s0 + s1;
                        sort_in_lds([kc](uint32_t i) { return kc[i]; }, [vc](uint32_t i) { return vc[i]; }, rg.x + s0 + s1, Lcc);
                        continue;
                    }
                    TileSortBufs h;
                    h.kA = kA + s0 + s1; h.vA = vA + s0 + s1; h.kB = kB + s0 + s1; h.vB = vB + s0 + s1;
                    finish_global(h, Lcc, sh2, rg.x + s0 + s1);
                }
            }
        }
    }  // slot loop
}

}  // namespace

size_t sort_hist_bytes(uint64_t D) {
    const uint64_t nblk = (D + GDR_SORT_TILE - 1) / GDR_SORT_TILE;
    return (size_t)((nblk > 0 ? nblk : 1) * GDR_RADIX + GDR_RADIX) * sizeof(uint32_t);
}

hipError_t launch_scan_block_sums(const gdr_geom* g, int N, hipStream_t st) {
    const int nb = div_up(N, GDR_BLOCK);
    GDR_LAUNCH(GDR_K_SCAN, scan_block_sums_kernel, dim3(1), dim3(GDR_BLOCK), st, g->block_sums, nb,
                       g->num_rendered);
    return hipGetLastError();
}

// ---- BinViews table of V views ------------------------------------------------------------------------------
void fill_bin_views(BinViews* vs, int V, const gdr_geom* geoms, const gdr_binning* bins, const gdr_image* imgs,
                    const uint64_t* D, const int32_t* const* radii) {
    memset(vs, 0, sizeof(*vs));
    for (int v = 0; v < V; ++v) {
        BinView& b = vs->v[v];
        const gdr_geom& g = geoms[v];
        const gdr_binning& bn = bins[v];
        b.radii = radii ? radii[v] : nullptr;
        b.depths = g.depths; b.rect = (const int4*)g.rect; b.tiles_touched = g.tiles_touched;
        b.block_offs = bn.global_sort ? g.block_sums : g.block_offs;
        b.keys[0] = bn.keys[0]; b.keys[1] = bn.keys[1]; b.vals[0] = bn.values[0]; b.vals[1] = bn.values[1];
        b.hist = bn.hist; b.scratch32 = bn.scratch32; b.tile_hist = bn.tile_hist; b.hist_width = bn.hist_width;
        b.from_totals = 0;
        b.ranges = (uint2*)imgs[v].ranges; b.tile_order = imgs[v].tile_order; b.seg_base = imgs[v].seg_base;
        b.seg_extra = (uint2*)bn.seg_extra; b.seg_count = bn.seg_count;
        b.D = D[v]; b.nblk = (uint32_t)((D[v] + GDR_SORT_TILE - 1) / GDR_SORT_TILE);
        b.d_dev = bn.d_dev;
        b.stats_out = bn.stats_out; b.hint_long = bn.hint_long; b.hint_medium = bn.hint_medium;
        b.seg_len = bn.seg_len; b.seg_cap = bn.seg_cap;
        b.deep_max_busy = (uint32_t)(bn.deep_max_busy > 0 ? bn.deep_max_busy : 0);
        b.deep_min_mean = (uint32_t)(bn.deep_min_mean > 0 ? bn.deep_min_mean : GDR_DEEP_MIN_MEAN);
    }
}

static uint64_t max_D(const BinViews& vs, int V) {
    uint64_t m = 0;
    for (int v = 0; v < V; ++v) m = vs.v[v].D > m ? vs.v[v].D : m;
    return m;
}

hipError_t launch_duplicate_views(const BinViews& vs, int V, int N, int W, int H, hipStream_t st) {
    const int tiles = tile_grid_x(W) * tile_grid_y(H);
    if (N == 0 || max_D(vs, V) == 0) {   // nothing to duplicate: only the ranges are cleared
        GDR_LAUNCH(GDR_K_RANGES, ranges_clear_kernel, dim3(div_up(tiles, GDR_BLOCK), V), dim3(GDR_BLOCK), st, vs, tiles);
        return hipGetLastError();
    }
    GDR_LAUNCH(GDR_K_DUPLICATE, duplicate_kernel, dim3(div_up(N, GDR_BLOCK), V), dim3(GDR_BLOCK), st, vs, N, tile_grid_x(W),
               tiles);
    return hipGetLastError();
}

// stable LSD passes over key bits [lo, hi) of every view; *sorted = buffer index that holds the result
hipError_t launch_sort_views(const BinViews& vs, int V, int lo, int hi, int* sorted, hipStream_t st) {
    *sorted = 0;
    const uint64_t md = max_D(vs, V);
    if (md == 0) return hipSuccess;
    const uint32_t nblk = (uint32_t)((md + GDR_SORT_TILE - 1) / GDR_SORT_TILE);
    int cur = 0;
    for (int shift = lo; shift < hi; shift += GDR_RADIX_BITS) {
        GDR_LAUNCH(GDR_K_SORT_HIST, sort_hist_kernel, dim3(nblk, V), dim3(GDR_BLOCK), st, vs, cur, shift);
        GDR_LAUNCH(GDR_K_SORT_ROWSCAN, sort_rowscan_kernel, dim3(GDR_RADIX, V), dim3(GDR_BLOCK), st, vs);
        if (shift == lo && hi - lo <= 2 * GDR_RADIX_BITS)   // first pass of the tile partition: evenly spread digits
            GDR_LAUNCH(GDR_K_SORT_SCATTER, sort_scatter_kernel<true>, dim3(nblk, V), dim3(GDR_BLOCK), st, vs, cur, shift);
        else
            GDR_LAUNCH(GDR_K_SORT_SCATTER, sort_scatter_kernel<false>, dim3(nblk, V), dim3(GDR_BLOCK), st, vs, cur, shift);
        cur ^= 1;
    }
    *sorted = cur;
    return hipGetLastError();
}

hipError_t launch_ranges_views(const BinViews& vs, int V, int cur, int tiles, hipStream_t st) {
    const uint64_t md = max_D(vs, V);   // (the ranges were cleared by launch_duplicate_views)
    if (md == 0) return hipSuccess;
    GDR_LAUNCH(GDR_K_RANGES, ranges_kernel, dim3(div_up((int64_t)md, GDR_BLOCK), V), dim3(GDR_BLOCK), st, vs, cur);
    return hipGetLastError();
}

// one launch covers all views: the largest hint of the views, 0 (= worst-case grid) if a view has none
static int merged_hint(const BinViews& vs, int V, bool long_class) {
    int h = 0;
    for (int v = 0; v < V; ++v) {
        const int x = long_class ? vs.v[v].hint_long : vs.v[v].hint_medium;
        if (x <= 0) return 0;
        h = x > h ? x : h;
    }
    return h;
}

// direct tile binning of V views (view = blockIdx.y; all views share N, the image size and hist_width):
// count -> scan -> [tile_order_kernel: totals -> ranges, issued by the caller] -> scatter
static void bin_geometry(const BinView& bv, int N, int tiles, int* nwg, int* chunk, int* tstride) {
    int w = div_up(N, GDR_BIN_THREADS);
    if (w > bv.hist_width) w = bv.hist_width;
    if (w < 1) w = 1;
    *chunk = div_up(div_up(N, w), GDR_BIN_THREADS) * GDR_BIN_THREADS;
    *nwg = div_up(N, *chunk);
    *tstride = div_up(tiles, 64) * 64;
}
hipError_t launch_tile_count_scan(const BinViews& vs, int V, int N, int W, int H, hipStream_t st) {
    const int gx = tile_grid_x(W), tiles = gx * tile_grid_y(H);
    int nwg, chunk, tstride;
    bin_geometry(vs.v[0], N, tiles, &nwg, &chunk, &tstride);
    const size_t lds = (size_t)tiles * sizeof(uint32_t);
    prof_begin(GDR_K_TILE_COUNT, st);
    hipLaunchKernelGGL(tile_count_kernel, dim3(nwg, V), dim3(GDR_BIN_THREADS), lds, st, vs, N, gx, tiles, chunk, tstride);
    prof_end(GDR_K_TILE_COUNT, st);
    GDR_LAUNCH(GDR_K_TILE_SCAN, tile_scan_kernel, dim3(div_up(tiles, 64), V), dim3(64 * GDR_BIN_SEGS), st, vs, tiles, nwg, tstride);
    return hipGetLastError();
}
hipError_t launch_tile_scatter(const BinViews& vs, int V, int N, int W, int H, hipStream_t st) {
    const int gx = tile_grid_x(W), tiles = gx * tile_grid_y(H);
    int nwg, chunk, tstride;
    bin_geometry(vs.v[0], N, tiles, &nwg, &chunk, &tstride);
    prof_begin(GDR_K_TILE_SCATTER, st);
    hipLaunchKernelGGL(tile_scatter_kernel, dim3(nwg, V), dim3(GDR_BIN_THREADS), (size_t)tiles * sizeof(uint32_t), st, vs, N, gx,
                       tiles, chunk, tstride);
    prof_end(GDR_K_TILE_SCATTER, st);
    return hipGetLastError();
}

// per-tile depth sort; input = buffers [in] (tile-partitioned; packed: one word per entry), output = values [in ^ 1]
// waves per workgroup of the medium class (measurement builds may change it: the forward-phase starvation study of round 6)
#ifndef GDR_TSORT_MEDIUM_WAVES
#define GDR_TSORT_MEDIUM_WAVES 8
#endif
#ifndef GDR_TSORT_NO_LONG
#define GDR_TSORT_NO_LONG 0
#endif
hipError_t launch_tile_sort_views(const BinViews& vs, int V, int in, int tiles, bool packed, hipStream_t st) {
    if (max_D(vs, V) == 0) return hipSuccess;
    // long lists: 16 waves per workgroup so that the few heavy tiles finish quickly; a two-class scheme (everything
    // beyond the short class bucketed and sorted in short-class chunks) measured slower at 2 M - 8 M Gaussians and equal
    // at 32 M; medium and long merged into one 16-wave class: C4 1251 -> 1231, C3 2924 -> 2897, shells +1 %; short and
    // medium merged into one class per tile (4 waves x 16 elements per lane, 36 KB): C3 shell +4.7 %, C2 -4.5 %; (8 waves,
    // 40 KB): C2 -2.3 %, others +-0
    // the long / medium classes are usually sparse or empty, and every one of their workgroups needs 144 / 40 KB of LDS on
    // a CU: with a hint from the previous call of this scene shape (gdr_binning.hint_*) only as many as there were tiles
    int g_long = tiles < 256 ? tiles : 256, g_medium = tiles < 512 ? tiles : 512;
    const int h_long = merged_hint(vs, V, true), h_medium = merged_hint(vs, V, false);
    if (h_long > 0 && h_long < g_long) g_long = h_long;
    if (h_medium > 0 && h_medium < g_medium) g_medium = h_medium;
    // hint_long < 0 (every view): the previous calls of this scene shape saw no list beyond the medium class.  The long
    // class is then not launched at all — a 16-wave workgroup with 144 KB of LDS needs a whole CU to itself, and in a
    // multi-view node, with other views' K6 workgroups resident everywhere, the EMPTY launch waited ~240 us for one
    // (kernel timeline of a C4 step, profiles/r03_timeline_c4.txt) and stalled its view's chain behind it.  The medium
    // class takes the lists beyond its LDS capacity instead (TOP: bucket pass through global memory, then LDS-sized
    // runs) — slower for such a list, identical result, so a wrong hint only costs time.
    bool no_long = true;
    for (int v = 0; v < V; ++v) no_long = no_long && vs.v[v].hint_long < 0;
    if (GDR_TSORT_NO_LONG) no_long = true;
    constexpr int MW = GDR_TSORT_MEDIUM_WAVES;
    if (packed) {
        if (!no_long)
            GDR_LAUNCH(GDR_K_TILE_SORT_LONG, (tile_sort_kernel<GDR_TSORT_LARGE, GDR_TSORT_MEDIUM, 16, true, true>),
                       dim3(g_long, V), dim3(16 * GDR_WAVE), st, vs, in, tiles);
        if (no_long)
            GDR_LAUNCH(GDR_K_TILE_SORT_LONG, (tile_sort_kernel<GDR_TSORT_MEDIUM, GDR_TSORT_SMALL, MW, true, true>),
                       dim3(g_medium, V), dim3(MW * GDR_WAVE), st, vs, in, tiles);
        else
            GDR_LAUNCH(GDR_K_TILE_SORT_LONG, (tile_sort_kernel<GDR_TSORT_MEDIUM, GDR_TSORT_SMALL, MW, false, true>),
                       dim3(g_medium, V), dim3(MW * GDR_WAVE), st, vs, in, tiles);
        GDR_LAUNCH(GDR_K_TILE_SORT, (tile_sort_kernel<GDR_TSORT_SMALL, 0, 4, false, true>), dim3(tiles, V), dim3(GDR_BLOCK), st,
                   vs, in, tiles);
        return hipGetLastError();
    }
    GDR_LAUNCH(GDR_K_TILE_SORT_LONG, (tile_sort_kernel<GDR_TSORT_LARGE, GDR_TSORT_MEDIUM, 16, true, false>),
               dim3(g_long, V), dim3(16 * GDR_WAVE), st, vs, in, tiles);
    GDR_LAUNCH(GDR_K_TILE_SORT_LONG, (tile_sort_kernel<GDR_TSORT_MEDIUM, GDR_TSORT_SMALL, 8, false, false>),
               dim3(g_medium, V), dim3(8 * GDR_WAVE), st, vs, in, tiles);
    GDR_LAUNCH(GDR_K_TILE_SORT, (tile_sort_kernel<GDR_TSORT_SMALL, 0, 4, false, false>), dim3(tiles, V), dim3(GDR_BLOCK), st, vs,
               in, tiles);
    return hipGetLastError();
}

}  // namespace gdr
