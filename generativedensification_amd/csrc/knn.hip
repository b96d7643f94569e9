// knn.hip — mean squared distance to the 3 nearest neighbours of every point: the `simple_knn._C.distCUDA2` the 2DGS
// adaptor imports (/root/reference/lightning/renderer_2dgs.py:11, used at :92-96 to initialise surfel scales; the only
// other use is point_decoder/layers/head.py:115).  SURVEY §8f-4.  The package is not in the reference tree; the
// published behaviour (graphdeco simple-knn): dist[i] = (d1 + d2 + d3) / 3 with d_k the squared distances of the three
// nearest OTHER points (by index: coincident points count with distance 0), fp32.
//
// Uniform grid instead of the lineage's Morton boxes: points are binned into G^3 cells (host plumbing: torch sort of the
// cell ids, bincount + cumsum for the cell starts), then one thread per point scans the cells of growing cubic shells
// around its own cell, keeping the three smallest distances, and stops as soon as the third best is closer than the
// nearest face of the cube searched so far (exact, not approximate).  HBM/L2-latency bound gather; ~27-125 cells of ~2
// points each per query at the default resolution.
#include "gdr_common.h"

namespace gdr {
namespace {

struct KnnGrid {
    const float* bbox;  // device: min x,y,z, max x,y,z
    int G;
};

__device__ __forceinline__ void grid_params(const KnnGrid& g, float* lo, float* inv_cs, float* cs) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = g.bbox[k];
        const float ext = fmaxf(g.bbox[3 + k] - g.bbox[k], 1e-20f);
        cs[k] = ext / (float)g.G;
        inv_cs[k] = (float)g.G / ext;
    }
}

__device__ __forceinline__ int cell_coord(float p, float lo, float inv_cs, int G) {
    return min(G - 1, max(0, (int)((p - lo) * inv_cs)));
}

__global__ __launch_bounds__(GDR_BLOCK) void knn_cells_kernel(const float* __restrict__ pts, int N, KnnGrid g,
                                                               int32_t* __restrict__ cell) {
    const int i = blockIdx.x * GDR_BLOCK + threadIdx.x;
    if (i >= N) return;
    float lo[3], ic[3], cs[3];
    grid_params(g, lo, ic, cs);
    const int cx = cell_coord(pts[3 * i], lo[0], ic[0], g.G), cy = cell_coord(pts[3 * i + 1], lo[1], ic[1], g.G);
    const int cz = cell_coord(pts[3 * i + 2], lo[2], ic[2], g.G);
    cell[i] = (cz * g.G + cy) * g.G + cx;
}

// pts: points in CELL-SORTED order; cell_start: (G^3 + 1) exclusive prefix of the cell populations; out[i] for sorted i
__global__ __launch_bounds__(GDR_BLOCK) void knn_mean_dist2_kernel(const float* __restrict__ pts, int N, KnnGrid g,
                                                                    const int32_t* __restrict__ cell_start,
                                                                    float* __restrict__ out) {
    const int i = blockIdx.x * GDR_BLOCK + threadIdx.x;
    if (i >= N) return;
    float lo[3], ic[3], cs[3];
    grid_params(g, lo, ic, cs);
    const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    const int G = g.G;
    const int cx = cell_coord(px, lo[0], ic[0], G), cy = cell_coord(py, lo[1], ic[1], G), cz = cell_coord(pz, lo[2], ic[2], G);
    float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
    auto visit = [&](int x, int y, int z) {
        const int c = (z * G + y) * G + x;
        const int e = cell_start[c + 1];
        for (int j = cell_start[c]; j < e; ++j) {
            if (j == i) continue;
            const float dx = pts[3 * j] - px, dy = pts[3 * j + 1] - py, dz = pts[3 * j + 2] - pz;
            float d = dx * dx + dy * dy + dz * dz;
            if (d < b0) { const float t = b0; b0 = d; d = t; }
            if (d < b1) { const float t = b1; b1 = d; d = t; }
            if (d < b2) b2 = d;
        }
    };
    for (int r = 0; r < G; ++r) {
        const int x0 = cx - r, x1 = cx + r, y0 = cy - r, y1 = cy + r, z0 = cz - r, z1 = cz + r;
        for (int z = max(z0, 0); z <= min(z1, G - 1); ++z)
            for (int y = max(y0, 0); y <= min(y1, G - 1); ++y) {
                const bool shell_zy = z == z0 || z == z1 || y == y0 || y == y1;
                if (shell_zy) {
                    for (int x = max(x0, 0); x <= min(x1, G - 1); ++x) visit(x, y, z);
                } else {  // interior rows of the shell: only the two end cells
                    if (x0 >= 0) visit(x0, y, z);
                    if (x1 < G && x1 != x0) visit(x1, y, z);
                }
            }
        // everything inside the cube [c - r, c + r]^3 has been seen: distance from the point to the cube's nearest
        // face that is still inside the grid bounds the distance of any unseen point from below
        float bound = INFINITY;
        if (x0 > 0) bound = fminf(bound, px - (lo[0] + (float)x0 * cs[0]));
        if (x1 < G - 1) bound = fminf(bound, (lo[0] + (float)(x1 + 1) * cs[0]) - px);
        if (y0 > 0) bound = fminf(bound, py - (lo[1] + (float)y0 * cs[1]));
        if (y1 < G - 1) bound = fminf(bound, (lo[1] + (float)(y1 + 1) * cs[1]) - py);
        if (z0 > 0) bound = fminf(bound, pz - (lo[2] + (float)z0 * cs[2]));
        if (z1 < G - 1) bound = fminf(bound, (lo[2] + (float)(z1 + 1) * cs[2]) - pz);
        if (bound == INFINITY) break;                       // the cube covers the whole grid
        bound = fmaxf(bound, 0.f) * 0.9999f;                // cell_coord clamps and rounds: stay conservative
        if (b2 <= bound * bound) break;
    }
    out[i] = (b0 + b1 + b2) / 3.0f;
}

}  // namespace

hipError_t launch_knn_cells(const float* pts, int N, const float* bbox, int G, int32_t* cell, hipStream_t st) {
    if (N == 0) return hipSuccess;
    const KnnGrid g{bbox, G};
    GDR_LAUNCH(GDR_K_KNN, knn_cells_kernel, dim3(div_up(N, GDR_BLOCK)), dim3(GDR_BLOCK), st, pts, N, g, cell);
    return hipGetLastError();
}

hipError_t launch_knn_mean_dist2(const float* pts_sorted, int N, const float* bbox, int G, const int32_t* cell_start,
                                 float* out, hipStream_t st) {
    if (N == 0) return hipSuccess;
    const KnnGrid g{bbox, G};
    GDR_LAUNCH(GDR_K_KNN, knn_mean_dist2_kernel, dim3(div_up(N, GDR_BLOCK)), dim3(GDR_BLOCK), st, pts_sorted, N, g,
               cell_start, out);
    return hipGetLastError();
}

}  // namespace gdr
