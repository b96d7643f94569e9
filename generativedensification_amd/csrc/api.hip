// api.hip — the C ABI of libgdr_hip.so (include/gdr.h): argument checking, workspace
// carving and stage sequencing.  Host code only; all kernels live in preprocess.hip,
// binning.hip and render.hip.  Nothing here allocates device memory or keeps state.
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "gdr_common.h"

namespace gdr {

static thread_local char g_err[512] = "";

void set_error(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s (%d)", what, e == hipSuccess ? "" : hipGetErrorString(e), (int)e);
}

// ---- opt-in per-kernel timing (process-wide; used by bench.py for the roofline) --------
struct ProfRec { int id; hipEvent_t a, b; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof_pending;
static std::vector<hipEvent_t> g_prof_pool;
static double g_prof_ms[GDR_K_COUNT];
static uint64_t g_prof_cnt[GDR_K_COUNT];
// the event opened by prof_begin and closed by the prof_end that follows it on the SAME host thread (GDR_LAUNCH
// brackets one launch): per thread, so that two host threads driving distinct workspaces with profiling on do not
// pair each other's events (the shared tables below are guarded by the mutex)
static thread_local hipEvent_t g_prof_open = nullptr;
static std::mutex g_prof_mu;

static hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void prof_begin(int id, hipStream_t st) {
    (void)id;
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_open = prof_event();
    (void)hipEventRecord(g_prof_open, st);
}
void prof_end(int id, hipStream_t st) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEvent_t b = prof_event();
    (void)hipEventRecord(b, st);
    g_prof_pending.push_back({id, g_prof_open, b});
    g_prof_open = nullptr;
}

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* b) : base((char*)b) {}
    template <typename T>
    T* take(size_t count) {
        T* p = base ? (T*)(base + off) : nullptr;
        off = align_up(off + count * sizeof(T));
        return p;
    }
};

static size_t carve_geom(void* base, int64_t N, gdr_geom* g) {
    Carver c(base);
    gdr_geom t;
    const size_t n = (size_t)(N > 0 ? N : 1);
    t.depths = c.take<float>(n);
    t.rec = c.take<float>(16 * n);
    t.cov3D = c.take<float>(6 * n);
    t.rect = c.take<int32_t>(4 * n);
    t.tiles_touched = c.take<uint32_t>(n);
    t.clamped = c.take<uint8_t>(n);
    t.block_sums = c.take<uint32_t>((n + GDR_BLOCK - 1) / GDR_BLOCK + 1);
    t.block_offs = c.take<uint32_t>((n + GDR_BLOCK - 1) / GDR_BLOCK + 1);
    t.num_rendered = c.take<uint32_t>(1);
    if (g) *g = t;
    return c.off;
}

static size_t carve_binning(void* base, uint64_t D, gdr_binning* b, int32_t seg_len = GDR_DEFAULT_SEG_LEN, int32_t N = 0,
                            int32_t tiles = 0) {
    Carver c(base);
    gdr_binning t;
    const size_t d = (size_t)(D > 0 ? D : 1);
    t.keys[0] = c.take<uint64_t>(d);
    t.keys[1] = c.take<uint64_t>(d);
    t.values[0] = c.take<uint32_t>(d);
    t.values[1] = c.take<uint32_t>(d);
    t.hist = c.take<uint32_t>(sort_hist_bytes(D) / sizeof(uint32_t));
    t.scratch32 = c.take<uint32_t>(2 * d);
    t.sorted = 0;
    t.global_sort = 0;
    t.deep_max_busy = GDR_DEFAULT_DEEP_MAX_BUSY;
    t.deep_min_mean = 0;
    t.d_dev = nullptr;
    t.stats_out = nullptr;
    t.hint_long = t.hint_medium = t.hint_no_deep = t.grad_rec_cleared = 0;
    t.seg_len = seg_len > 0 ? (seg_len + GDR_BLOCK - 1) / GDR_BLOCK * GDR_BLOCK : 0;  // callers may raise it (a multiple of 256) or set 0 after carving (include/gdr.h)
    t.seg_cap = t.seg_len ? (int32_t)(D / (uint64_t)t.seg_len + 1) : 0;
    t.seg_extra = c.take<uint32_t>(2 * (size_t)(t.seg_cap ? t.seg_cap : 1));
    t.seg_count = c.take<uint32_t>(4);
    t.seg_state = c.take<float>(t.seg_cap ? (size_t)2 * t.seg_cap * GDR_SEG_STATE_FLOATS : 1);
    t.tile_hist = nullptr;
    t.hist_width = 0;
    t.reserved1 = 0;
    if (N > 0 && tiles > 0 && tiles <= GDR_BIN_MAX_TILES) {   // direct tile binning: (width rows x tiles) counts + a totals row
        int w = (N + 1023) / 1024;
        t.hist_width = w > GDR_BIN_MAX_WIDTH ? GDR_BIN_MAX_WIDTH : w;
        t.tile_hist = c.take<uint32_t>((size_t)(t.hist_width + 1) * ((tiles + 63) / 64 * 64));
    }
    if (b) *b = t;
    return c.off;
}

static size_t carve_image(void* base, int H, int W, gdr_image* im) {
    Carver c(base);
    gdr_image t;
    const size_t tiles = (size_t)tile_grid_x(W) * tile_grid_y(H);
    const size_t P = (size_t)H * W;
    t.ranges = c.take<uint32_t>(2 * (tiles ? tiles : 1));
    t.n_contrib = c.take<uint32_t>(P ? P : 1);
    t.final_T = c.take<float>(P ? P : 1);
    t.tile_order = c.take<uint32_t>(tiles ? tiles : 1);
    t.seg_base = c.take<uint32_t>(tiles ? tiles : 1);
    if (im) *im = t;
    return c.off;
}

static int check_common(const gdr_settings* s, const gdr_inputs* in) {
    if (!s || !in) { set_error("NULL settings/inputs", hipSuccess); return GDR_ERR_INVALID_ARG; }
    if (in->N < 0 || s->image_height <= 0 || s->image_width <= 0) {
        set_error("negative N or empty image", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    if (in->N > GDR_MAX_GAUSSIANS) { set_error("N exceeds GDR_MAX_GAUSSIANS", hipSuccess); return GDR_ERR_UNSUPPORTED; }
    if (!s->bg || !s->viewmatrix || !s->projmatrix) {
        set_error("bg/viewmatrix/projmatrix must be device pointers", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    if (in->N > 0) {
        if (!in->means3D || !in->opacities) { set_error("means3D/opacities NULL", hipSuccess); return GDR_ERR_INVALID_ARG; }
        if ((in->shs != nullptr) == (in->colors_precomp != nullptr)) {
            set_error("provide exactly one of shs / colors_precomp", hipSuccess);
            return GDR_ERR_INVALID_ARG;
        }
        const bool sr = in->scales && in->rotations;
        if (sr == (in->cov3D_precomp != nullptr) || ((in->scales != nullptr) != (in->rotations != nullptr))) {
            set_error("provide exactly one of (scales, rotations) / cov3D_precomp", hipSuccess);
            return GDR_ERR_INVALID_ARG;
        }
        if (in->shs) {
            if (s->sh_degree < 0 || s->sh_degree > 3) { set_error("sh_degree must be 0..3", hipSuccess); return GDR_ERR_UNSUPPORTED; }
            if (in->M < (s->sh_degree + 1) * (s->sh_degree + 1)) { set_error("M < (sh_degree+1)^2", hipSuccess); return GDR_ERR_INVALID_ARG; }
            if (!s->campos) { set_error("campos NULL", hipSuccess); return GDR_ERR_INVALID_ARG; }
        }
    }
    return GDR_OK;
}

static int hip_fail(const char* what, hipError_t e) {
    set_error(what, e);
    return GDR_ERR_HIP;
}

static int debug_sync(const gdr_settings* s, const char* what, hipStream_t st) {
    if (!s->debug) return GDR_OK;
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) return hip_fail(what, e);
    return GDR_OK;
}

}  // namespace gdr

using namespace gdr;

extern "C" {

int gdr_abi_version(void) { return GDR_ABI_VERSION; }
#ifndef GDR_BUILD_TAG
#define GDR_BUILD_TAG "release"
#endif
const char* gdr_build_tag(void) { return GDR_BUILD_TAG; }
const char* gdr_last_error(void) { return g_err; }

size_t gdr_geom_bytes(int32_t N) { return carve_geom(nullptr, N, nullptr); }
size_t gdr_binning_bytes(uint64_t D) { return carve_binning(nullptr, D, nullptr); }
size_t gdr_image_bytes(int32_t H, int32_t W) { return carve_image(nullptr, H, W, nullptr); }

int gdr_geom_carve(void* base, int32_t N, gdr_geom* out) {
    if (!base || !out || ((uintptr_t)base & 255u)) { set_error("geom base NULL/unaligned", hipSuccess); return GDR_ERR_INVALID_ARG; }
    carve_geom(base, N, out);
    return GDR_OK;
}
int gdr_binning_carve(void* base, uint64_t D, gdr_binning* out) {
    if (!base || !out || ((uintptr_t)base & 255u)) { set_error("binning base NULL/unaligned", hipSuccess); return GDR_ERR_INVALID_ARG; }
    carve_binning(base, D, out);
    return GDR_OK;
}
size_t gdr_binning_bytes_for(uint64_t D, int32_t seg_len, int32_t N, int32_t tiles) {
    return carve_binning(nullptr, D, nullptr, seg_len, N, tiles);
}
int gdr_binning_carve_for(void* base, uint64_t D, int32_t seg_len, int32_t N, int32_t tiles, gdr_binning* out) {
    if (!base || !out || ((uintptr_t)base & 255u) || seg_len < 0 || N < 0 || tiles < 0) {
        set_error("binning base NULL/unaligned or negative seg_len / N / tiles", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    carve_binning(base, D, out, seg_len, N, tiles);
    return GDR_OK;
}
int gdr_image_carve(void* base, int32_t H, int32_t W, gdr_image* out) {
    if (!base || !out || ((uintptr_t)base & 255u)) { set_error("image base NULL/unaligned", hipSuccess); return GDR_ERR_INVALID_ARG; }
    carve_image(base, H, W, out);
    return GDR_OK;
}

int gdr_preprocess_forward(const gdr_settings* s, const gdr_inputs* in, const gdr_geom* geom,
                           int32_t* radii, uint32_t* num_rendered_host, void* stream) {
    int rc = check_common(s, in);
    if (rc) return rc;
    if (!geom || (in->N > 0 && !radii)) { set_error("geom/radii NULL", hipSuccess); return GDR_ERR_INVALID_ARG; }
    const int tiles = tile_grid_x(s->image_width) * tile_grid_y(s->image_height);
    if (key_bits(tiles) > 64) { set_error("image too large", hipSuccess); return GDR_ERR_UNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(geom->num_rendered, 0, sizeof(uint32_t), st);  // K1 draws block offsets from it
    if (e != hipSuccess) return hip_fail("memset num_rendered", e);
    e = launch_preprocess_fwd(s, in, geom, radii, st);
    if (e != hipSuccess) return hip_fail("preprocess_fwd", e);
    if ((rc = debug_sync(s, "preprocess_fwd", st))) return rc;
    if (num_rendered_host) {
        e = hipMemcpyAsync(num_rendered_host, geom->num_rendered, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return hip_fail("memcpy num_rendered", e);
        e = hipStreamSynchronize(st);
        if (e != hipSuccess) return hip_fail("sync num_rendered", e);
    }
    return GDR_OK;
}

// Everything between K1 and K6 for V views (every launch covers all of them: view = blockIdx.y); shared by the 3DGS and the
// surfel path (only the geometry's depths / rects / tiles_touched and the radii are read).  All views share one image
// size, N and workspace shape.  Default: direct tile binning (count / scan / scatter) -> tile order -> per-tile LDS depth
// sort.  Without a count matrix (gdr_binning_carve, or > 16384 tiles): duplicate + stable radix partition on the tile
// bits -> ranges -> the same tile order and sort.  global_sort: one stable LSD radix sort over all key bits.
static int binning_stage_views(int V, const gdr_settings* s, int32_t N, const gdr_geom* geoms, gdr_binning* bins,
                               const gdr_image* imgs, const uint64_t* D, const int32_t* const* radii, hipStream_t st) {
    int rc;
    const int W = s->image_width, H = s->image_height;
    const int tiles = tile_grid_x(W) * tile_grid_y(H);
    hipError_t e;
    const bool global_sort = bins[0].global_sort != 0;
    bool direct = !global_sort && tiles <= GDR_BIN_MAX_TILES;
    uint64_t dmax = 0;
    for (int v = 0; v < V; ++v) {
        direct = direct && bins[v].tile_hist && bins[v].hist_width > 0 && bins[v].hist_width == bins[0].hist_width;
        dmax = D[v] > dmax ? D[v] : dmax;
    }
    if (global_sort) {  // stable global sort: duplicates must be emitted in Gaussian order
        for (int v = 0; v < V; ++v) {
            e = launch_scan_block_sums(&geoms[v], N, st);
            if (e != hipSuccess) return hip_fail("scan_block_sums", e);
        }
    }
    BinViews vs;
    fill_bin_views(&vs, V, geoms, bins, imgs, D, radii);
    int sorted = 0;
    bool from_totals = false;
    if (direct && (N == 0 || dmax == 0)) {   // nothing to bin: only the ranges are cleared (ranges_clear inside)
        e = launch_duplicate_views(vs, V, 0, W, H, st);
        if (e != hipSuccess) return hip_fail("ranges_clear", e);
    } else if (direct) {
        e = launch_tile_count_scan(vs, V, N, W, H, st);
        if (e != hipSuccess) return hip_fail("tile_count_scan", e);
        if ((rc = debug_sync(s, "tile_count_scan", st))) return rc;
        from_totals = true;
        for (int v = 0; v < V; ++v) vs.v[v].from_totals = 1;
    } else {
        e = launch_duplicate_views(vs, V, N, W, H, st);
        if (e != hipSuccess) return hip_fail("duplicate", e);
        if ((rc = debug_sync(s, "duplicate", st))) return rc;
        e = launch_sort_views(vs, V, global_sort ? 0 : 32, key_bits(tiles), &sorted, st);
        if (e != hipSuccess) return hip_fail("sort", e);
        if ((rc = debug_sync(s, "sort", st))) return rc;
        e = launch_ranges_views(vs, V, sorted, tiles, st);
        if (e != hipSuccess) return hip_fail("ranges", e);
        if ((rc = debug_sync(s, "ranges", st))) return rc;
    }
    e = launch_tile_order_views(vs, V, tiles, st);  // (totals -> ranges first;) longest list first: launch order of tile_sort, K6, K7
    if (e != hipSuccess) return hip_fail("tile_order", e);
    if ((rc = debug_sync(s, "tile_order", st))) return rc;
    if (from_totals) {
        e = launch_tile_scatter(vs, V, N, W, H, st);
        if (e != hipSuccess) return hip_fail("tile_scatter", e);
        if ((rc = debug_sync(s, "tile_scatter", st))) return rc;
    }
    if (!global_sort) {  // per-tile LDS depth sort of the partitioned lists
        e = launch_tile_sort_views(vs, V, sorted, tiles, direct, st);
        if (e != hipSuccess) return hip_fail("tile_sort", e);
        if ((rc = debug_sync(s, "tile_sort", st))) return rc;
        sorted ^= 1;
    }
    for (int v = 0; v < V; ++v) bins[v].sorted = sorted;
    return GDR_OK;
}

static int binning_stage(const gdr_settings* s, int32_t N, const gdr_geom* geom, gdr_binning* bin, const gdr_image* img,
                         uint64_t D, const int32_t* radii, hipStream_t st) {
    return binning_stage_views(1, s, N, geom, bin, img, &D, &radii, st);
}

int gdr_binning_forward(const gdr_settings* s, int32_t N, const gdr_geom* geom, gdr_binning* bin, const gdr_image* img,
                        uint64_t D, const int32_t* radii, void* stream) {
    if (!s || !geom || !bin || !img || N < 0 || (N > 0 && !radii)) { set_error("binning_forward: bad argument", hipSuccess); return GDR_ERR_INVALID_ARG; }
    return binning_stage(s, N, geom, bin, img, D, radii, (hipStream_t)stream);
}

int gdr_composite_forward(const gdr_settings* s, const gdr_geom* geom, const gdr_binning* bin, const gdr_image* img,
                          const gdr_outputs* out, void* stream) {
    if (!s || !geom || !bin || !img || !out || !out->color || !out->depth || !out->alpha) {
        set_error("composite_forward: NULL argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = launch_render_fwd(s, geom, bin, img, out, st);
    if (e != hipSuccess) return hip_fail("render_fwd", e);
    return debug_sync(s, "render_fwd", st);
}

int gdr_composite_forward_loss(const gdr_settings* s, const gdr_geom* geom, const gdr_binning* bin, const gdr_image* img,
                               const gdr_outputs* out, const float* target, float w_depth, float w_alpha, float* loss,
                               void* stream) {
    if (!s || !geom || !bin || !img || !out || !out->color || !out->depth || !out->alpha || !target || !loss) {
        set_error("composite_forward_loss: NULL argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = launch_render_fwd_loss(s, geom, bin, img, out, target, w_depth, w_alpha, loss, st);
    if (e != hipSuccess) return hip_fail("render_fwd_loss", e);
    return debug_sync(s, "render_fwd_loss", st);
}

int gdr_composite_forward_lossgrad(const gdr_settings* s, const gdr_geom* geom, const gdr_binning* bin, const gdr_image* img,
                                   const float* target, float go_scale, float* loss, float* dL_dcolor, void* stream) {
    if (!s || !geom || !bin || !img || !target || !loss || !dL_dcolor) {
        set_error("composite_forward_lossgrad: NULL argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = launch_render_fwd_lossgrad(s, geom, bin, img, target, go_scale, loss, dL_dcolor, st);
    if (e != hipSuccess) return hip_fail("render_fwd_lossgrad", e);
    return debug_sync(s, "render_fwd_lossgrad", st);
}

int gdr_words_differ(const void* a, const void* b, uint64_t n_bytes, uint32_t* flag, void* stream) {
    if (!flag || (n_bytes && (!a || !b)) || (n_bytes & 3u) || (((uintptr_t)a | (uintptr_t)b) & 15u)) {
        set_error("words_differ: NULL / unaligned argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    if (n_bytes == 0) return GDR_OK;
    hipError_t e = launch_words_differ(a, b, n_bytes, flag, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail("words_differ", e);
    return GDR_OK;
}

// ---- host-boundary helper: a small device -> pinned-host copy with its own pooled event (the duplicate count of a call) ----
namespace {
struct HostCopyTicket { hipEvent_t ev; int dev; };
std::mutex g_ticket_mu;
std::vector<HostCopyTicket*> g_ticket_pool;
}  // namespace

int gdr_host_copy_begin(void* dst_pinned, const void* src_dev, uint64_t n_bytes, void* stream, void** ticket) {
    if (!dst_pinned || !src_dev || !ticket || n_bytes == 0) { set_error("host_copy_begin: NULL argument", hipSuccess); return GDR_ERR_INVALID_ARG; }
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail("host_copy_begin: hipGetDevice", e);
    HostCopyTicket* t = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_ticket_mu);
        for (size_t k = 0; k < g_ticket_pool.size(); ++k)
            if (g_ticket_pool[k]->dev == dev) { t = g_ticket_pool[k]; g_ticket_pool[k] = g_ticket_pool.back(); g_ticket_pool.pop_back(); break; }
    }
    if (!t) {
        t = new HostCopyTicket{nullptr, dev};
        e = hipEventCreateWithFlags(&t->ev, hipEventDisableTiming);
        if (e != hipSuccess) { delete t; return hip_fail("host_copy_begin: hipEventCreate", e); }
    }
    e = hipMemcpyAsync(dst_pinned, src_dev, n_bytes, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipEventRecord(t->ev, (hipStream_t)stream);
    if (e != hipSuccess) {
        std::lock_guard<std::mutex> lk(g_ticket_mu);
        g_ticket_pool.push_back(t);
        return hip_fail("host_copy_begin", e);
    }
    *ticket = t;
    return GDR_OK;
}

int gdr_host_copy_wait(void* ticket) {
    if (!ticket) { set_error("host_copy_wait: NULL ticket", hipSuccess); return GDR_ERR_INVALID_ARG; }
    HostCopyTicket* t = (HostCopyTicket*)ticket;
    const hipError_t e = hipEventSynchronize(t->ev);
    {
        std::lock_guard<std::mutex> lk(g_ticket_mu);
        g_ticket_pool.push_back(t);
    }
    if (e != hipSuccess) return hip_fail("host_copy_wait", e);
    return GDR_OK;
}

int gdr_words_differ_multi(int32_t n, const void* const* a, const void* const* b, const uint64_t* n_bytes, uint32_t* flag,
                           void* stream) {
    if (n < 0 || n > GDR_DIFFER_MAX || !flag || (n && (!a || !b || !n_bytes))) {
        set_error("words_differ_multi: bad argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    for (int k = 0; k < n; ++k)
        if ((n_bytes[k] && (!a[k] || !b[k])) || (n_bytes[k] & 3u) || (((uintptr_t)a[k] | (uintptr_t)b[k]) & 15u)) {
            set_error("words_differ_multi: NULL / unaligned buffer", hipSuccess);
            return GDR_ERR_INVALID_ARG;
        }
    if (n == 0) return GDR_OK;
    hipError_t e = launch_words_differ_multi(n, a, b, n_bytes, flag, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail("words_differ_multi", e);
    return GDR_OK;
}

int gdr_clear_async(void* dst, uint64_t n_bytes, void* stream) {
    if (n_bytes == 0) return GDR_OK;
    if (!dst) { set_error("clear_async: NULL argument", hipSuccess); return GDR_ERR_INVALID_ARG; }
    const hipError_t e = hipMemsetAsync(dst, 0, (size_t)n_bytes, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail("clear_async", e);
    return GDR_OK;
}

size_t gdr_topk_workspace_bytes(void) { return select_workspace_bytes(); }

int gdr_topk_absgrad(int32_t N, const float* grad, const uint8_t* candidates, int32_t k, void* workspace, uint8_t* mask,
                     int32_t* indices, void* stream) {
    if (N < 0 || k < 0 || (N > 0 && (!grad || !workspace || !mask))) { set_error("topk_absgrad: bad argument", hipSuccess); return GDR_ERR_INVALID_ARG; }
    if (N == 0) return GDR_OK;
    hipError_t e = launch_topk_absgrad(N, grad, candidates, k, workspace, mask, indices, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail("topk_absgrad", e);
    return GDR_OK;
}

int gdr_render_backward_loss(const gdr_settings* s, int32_t N, const gdr_geom* geom, const gdr_binning* bin,
                             const gdr_image* img, const float* color, const float* target, float w_depth, float w_alpha,
                             const float* g, float* grad_rec, void* stream) {
    if (!s || !geom || !bin || !img || !color || !target || !g || (N > 0 && !grad_rec) || !s->bg) {
        set_error("render_backward_loss: NULL argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    if (N <= 0) return GDR_OK;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = bin->grad_rec_cleared ? hipSuccess : hipMemsetAsync(grad_rec, 0, (size_t)N * 16 * sizeof(float), st);
    if (e != hipSuccess) return hip_fail("memset gradient records", e);
    e = launch_render_bwd_loss(s, geom, bin, img, color, target, w_depth, w_alpha, g, grad_rec, st);
    if (e != hipSuccess) return hip_fail("render_bwd_loss", e);
    return debug_sync(s, "render_bwd_loss", st);
}

int gdr_render_backward_mean2d_loss(const gdr_settings* s, int32_t N, const gdr_geom* geom, const gdr_binning* bin,
                                    const gdr_image* img, const float* color, const float* target, const float* g,
                                    float* dL_dmean2D, void* stream) {
    if (!s || !geom || !bin || !img || !color || !target || !g || (N > 0 && !dL_dmean2D) || !s->bg) {
        set_error("render_backward_mean2d_loss: NULL argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    if (N <= 0) return GDR_OK;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = launch_render_bwd_mean2d_loss(s, geom, bin, img, color, target, g, dL_dmean2D, st);
    if (e != hipSuccess) return hip_fail("render_bwd_mean2d_loss", e);
    return debug_sync(s, "render_bwd_mean2d_loss", st);
}

int gdr_render_forward(const gdr_settings* s, const gdr_inputs* in, const gdr_geom* geom,
                       gdr_binning* bin, const gdr_image* img, uint64_t D, const gdr_outputs* out,
                       void* stream) {
    int rc = check_common(s, in);
    if (rc) return rc;
    if (!geom || !bin || !img || !out || !out->color || !out->depth || !out->alpha || (in->N > 0 && !out->radii)) {
        set_error("render_forward: NULL argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    if ((rc = binning_stage(s, in->N, geom, bin, img, D, out->radii, (hipStream_t)stream))) return rc;
    return gdr_composite_forward(s, geom, bin, img, out, stream);
}

int gdr_forward(const gdr_settings* s, const gdr_inputs* in, const gdr_geom* geom, gdr_binning* bin,
                const gdr_image* img, uint64_t D_cap, const gdr_outputs* out,
                uint32_t* num_rendered_host, void* stream) {
    if (!out || !num_rendered_host) { set_error("gdr_forward: NULL out/num_rendered_host", hipSuccess); return GDR_ERR_INVALID_ARG; }
    int rc = gdr_preprocess_forward(s, in, geom, out->radii, num_rendered_host, stream);
    if (rc) return rc;
    if ((uint64_t)*num_rendered_host > D_cap) {
        set_error("binning workspace too small for num_rendered", hipSuccess);
        return GDR_ERR_WORKSPACE;
    }
    return gdr_render_forward(s, in, geom, bin, img, *num_rendered_host, out, stream);
}

int gdr_backward(const gdr_settings* s, const gdr_inputs* in, const gdr_geom* geom,
                 const gdr_binning* bin, const gdr_image* img, uint64_t D, const int32_t* radii,
                 const gdr_grad_inputs* gin, const gdr_grad_outputs* gout, void* stream) {
    (void)D;
    int rc = check_common(s, in);
    if (rc) return rc;
    if (in->N == 0) return GDR_OK;  // no Gaussians: the (empty) gradient buffers may be NULL
    if (!geom || !bin || !img || !gin || !gout || !gin->dL_dcolor || !gout->dL_dmeans3D ||
        !gout->dL_dmeans2D || !gout->dL_dopacities || !gout->scratch || (in->N > 0 && !radii)) {
        set_error("backward: NULL argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    if (in->shs && !gout->dL_dshs) { set_error("backward: dL_dshs NULL", hipSuccess); return GDR_ERR_INVALID_ARG; }
    if (in->colors_precomp && !gout->dL_dcolors) { set_error("backward: dL_dcolors NULL", hipSuccess); return GDR_ERR_INVALID_ARG; }
    if (in->cov3D_precomp ? !gout->dL_dcov3D : (!gout->dL_dscales || !gout->dL_drotations)) {
        set_error("backward: covariance gradient buffers NULL", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t N = (size_t)in->N;
    if (N == 0) return GDR_OK;
    hipError_t e = hipMemsetAsync(gout->scratch, 0, N * 16 * sizeof(float), st);
    if (e != hipSuccess) return hip_fail("memset gradient records", e);
    e = launch_render_bwd(s, geom, bin, img, gin, gout, st);
    if (e != hipSuccess) return hip_fail("render_bwd", e);
    if ((rc = debug_sync(s, "render_bwd", st))) return rc;
    e = launch_preprocess_bwd(s, in, geom, radii, gout, st);
    if (e != hipSuccess) return hip_fail("preprocess_bwd", e);
    if ((rc = debug_sync(s, "preprocess_bwd", st))) return rc;
    return GDR_OK;
}

int gdr_preprocess_forward_views(int32_t V, const gdr_settings* s, const gdr_inputs* in,
                                 const gdr_geom* geoms, int32_t* const* radii, void* stream) {
    if (V < 1 || V > GDR_MAX_VIEWS) { set_error("views: V out of range", hipSuccess); return GDR_ERR_UNSUPPORTED; }
    if (!s || !in || !geoms || !radii) { set_error("views: NULL argument", hipSuccess); return GDR_ERR_INVALID_ARG; }
    for (int v = 0; v < V; ++v) {
        int rc = check_common(&s[v], in);
        if (rc) return rc;
        if (s[v].image_width != s[0].image_width || s[v].image_height != s[0].image_height ||
            s[v].sh_degree != s[0].sh_degree || s[v].scale_modifier != s[0].scale_modifier) {
            set_error("views: image size / sh_degree / scale_modifier must match", hipSuccess);
            return GDR_ERR_INVALID_ARG;
        }
        if (in->N > 0 && !radii[v]) { set_error("views: radii NULL", hipSuccess); return GDR_ERR_INVALID_ARG; }
    }
    if (in->N > 0 && (!in->shs || !in->scales || !in->rotations)) {
        set_error("views: needs shs + scales + rotations", hipSuccess);
        return GDR_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    bool packed = true;  // the V counters in one array (geoms[v].num_rendered = base + v): one fill instead of V
    for (int v = 1; v < V; ++v) packed = packed && geoms[v].num_rendered == geoms[0].num_rendered + v;
    if (packed) e = hipMemsetAsync(geoms[0].num_rendered, 0, (size_t)V * sizeof(uint32_t), st);
    else
        for (int v = 0; v < V && e == hipSuccess; ++v) e = hipMemsetAsync(geoms[v].num_rendered, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return hip_fail("memset num_rendered", e);
    e = launch_preprocess_fwd_views(V, s, in, geoms, radii, st);
    if (e != hipSuccess) return hip_fail("preprocess_fwd_views", e);
    return debug_sync(&s[0], "preprocess_fwd_views", st);
}

int gdr_render_backward(const gdr_settings* s, int32_t N, const gdr_geom* geom, const gdr_binning* bin,
                        const gdr_image* img, const gdr_grad_inputs* gin, float* grad_rec,
                        void* stream) {
    if (!s || !geom || !bin || !img || !gin || !gin->dL_dcolor || (N > 0 && !grad_rec) || !s->bg) {
        set_error("render_backward: NULL argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    if (N <= 0) return GDR_OK;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = bin->grad_rec_cleared ? hipSuccess : hipMemsetAsync(grad_rec, 0, (size_t)N * 16 * sizeof(float), st);
    if (e != hipSuccess) return hip_fail("memset gradient records", e);
    gdr_grad_outputs go;
    memset(&go, 0, sizeof(go));
    go.scratch = grad_rec;
    e = launch_render_bwd(s, geom, bin, img, gin, &go, st);
    if (e != hipSuccess) return hip_fail("render_bwd", e);
    return debug_sync(s, "render_bwd", st);
}

// K7 of V views in one launch (round 4).  Every view's record is cleared here unless its bins[v].grad_rec_cleared says
// the caller did.
static int check_bwd_views(int32_t V, const gdr_settings* s, const gdr_geom* geoms, const gdr_binning* bins,
                           const gdr_image* imgs, const char* what) {
    if (V < 1 || V > GDR_MAX_VIEWS) { set_error("views: V out of range", hipSuccess); return GDR_ERR_UNSUPPORTED; }
    if (!s || !geoms || !bins || !imgs) { set_error(what, hipSuccess); return GDR_ERR_INVALID_ARG; }
    for (int v = 0; v < V; ++v)
        if (!s[v].bg || s[v].image_width != s[0].image_width || s[v].image_height != s[0].image_height) {
            set_error("views: bg NULL or image sizes differ", hipSuccess);
            return GDR_ERR_INVALID_ARG;
        }
    return GDR_OK;
}

int gdr_render_backward_views(int32_t V, const gdr_settings* s, int32_t N, const gdr_geom* geoms, const gdr_binning* bins,
                              const gdr_image* imgs, const gdr_grad_inputs* gins, float* const* grad_recs,
                              int32_t interleave, void* stream) {
    int rc = check_bwd_views(V, s, geoms, bins, imgs, "render_backward_views: NULL argument");
    if (rc) return rc;
    if (N <= 0) return GDR_OK;
    if (!gins || !grad_recs) { set_error("render_backward_views: NULL argument", hipSuccess); return GDR_ERR_INVALID_ARG; }
    hipStream_t st = (hipStream_t)stream;
    for (int v = 0; v < V; ++v) {
        if (!gins[v].dL_dcolor || !grad_recs[v]) { set_error("render_backward_views: NULL view buffer", hipSuccess); return GDR_ERR_INVALID_ARG; }
        if (!bins[v].grad_rec_cleared) {
            hipError_t e = hipMemsetAsync(grad_recs[v], 0, (size_t)N * 16 * sizeof(float), st);
            if (e != hipSuccess) return hip_fail("memset gradient records", e);
        }
    }
    hipError_t e = launch_render_bwd_views(V, s, geoms, bins, imgs, gins, grad_recs, interleave, st);
    if (e != hipSuccess) return hip_fail("render_bwd_views", e);
    return debug_sync(&s[0], "render_bwd_views", st);
}

int gdr_render_backward_loss_views(int32_t V, const gdr_settings* s, int32_t N, const gdr_geom* geoms,
                                   const gdr_binning* bins, const gdr_image* imgs, const float* const* colors,
                                   const float* const* targets, float w_depth, float w_alpha, const float* g,
                                   float* const* grad_recs, int32_t interleave, void* stream) {
    int rc = check_bwd_views(V, s, geoms, bins, imgs, "render_backward_loss_views: NULL argument");
    if (rc) return rc;
    if (N <= 0) return GDR_OK;
    if (!colors || !targets || !g || !grad_recs) { set_error("render_backward_loss_views: NULL argument", hipSuccess); return GDR_ERR_INVALID_ARG; }
    hipStream_t st = (hipStream_t)stream;
    for (int v = 0; v < V; ++v) {
        if (!colors[v] || !targets[v] || !grad_recs[v]) { set_error("render_backward_loss_views: NULL view buffer", hipSuccess); return GDR_ERR_INVALID_ARG; }
        if (!bins[v].grad_rec_cleared) {
            hipError_t e = hipMemsetAsync(grad_recs[v], 0, (size_t)N * 16 * sizeof(float), st);
            if (e != hipSuccess) return hip_fail("memset gradient records", e);
        }
    }
    hipError_t e = launch_render_bwd_loss_views(V, s, geoms, bins, imgs, colors, targets, w_depth, w_alpha, g, grad_recs,
                                                interleave, st);
    if (e != hipSuccess) return hip_fail("render_bwd_loss_views", e);
    return debug_sync(&s[0], "render_bwd_loss_views", st);
}

int gdr_render_backward_mean2d_views(int32_t V, const gdr_settings* s, int32_t N, const gdr_geom* geoms,
                                     const gdr_binning* bins, const gdr_image* imgs, const float* const* dL_dcolors,
                                     float* dL_dmean2D, int32_t interleave, void* stream) {
    int rc = check_bwd_views(V, s, geoms, bins, imgs, "render_backward_mean2d_views: NULL argument");
    if (rc) return rc;
    if (N <= 0) return GDR_OK;
    if (!dL_dcolors || !dL_dmean2D) { set_error("render_backward_mean2d_views: NULL argument", hipSuccess); return GDR_ERR_INVALID_ARG; }
    for (int v = 0; v < V; ++v)
        if (!dL_dcolors[v]) { set_error("render_backward_mean2d_views: NULL view buffer", hipSuccess); return GDR_ERR_INVALID_ARG; }
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = launch_render_bwd_mean2d_views(V, s, geoms, bins, imgs, dL_dcolors, dL_dmean2D, interleave, st);
    if (e != hipSuccess) return hip_fail("render_bwd_mean2d_views", e);
    return debug_sync(&s[0], "render_bwd_mean2d_views", st);
}

int gdr_render_backward_mean2d(const gdr_settings* s, int32_t N, const gdr_geom* geom,
                               const gdr_binning* bin, const gdr_image* img, const float* dL_dcolor,
                               float* dL_dmean2D, void* stream) {
    if (!s || !geom || !bin || !img || !dL_dcolor || (N > 0 && !dL_dmean2D) || !s->bg) {
        set_error("render_backward_mean2d: NULL argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    if (N <= 0) return GDR_OK;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = launch_render_bwd_mean2d(s, geom, bin, img, dL_dcolor, dL_dmean2D, st);
    if (e != hipSuccess) return hip_fail("render_bwd_mean2d", e);
    return debug_sync(s, "render_bwd_mean2d", st);
}

int gdr_preprocess_backward_views(int32_t V, const gdr_settings* s, const gdr_inputs* in,
                                  const gdr_geom* geoms, const int32_t* const* radii,
                                  float* const* grad_recs, const gdr_grad_outputs* gout, void* stream) {
    if (V < 1 || V > GDR_MAX_VIEWS) { set_error("views: V out of range", hipSuccess); return GDR_ERR_UNSUPPORTED; }
    if (s && in && in->N == 0) return GDR_OK;  // no Gaussians: the (empty) gradient buffers may be NULL
    if (!s || !in || !geoms || !radii || !grad_recs || !gout || !gout->dL_dmeans3D || !gout->dL_dmeans2D ||
        !gout->dL_dshs || !gout->dL_dopacities || !gout->dL_dscales || !gout->dL_drotations) {
        set_error("backward_views: NULL argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    for (int v = 0; v < V; ++v) {
        int rc = check_common(&s[v], in);
        if (rc) return rc;
        if (in->N > 0 && (!radii[v] || !grad_recs[v])) { set_error("backward_views: NULL view buffer", hipSuccess); return GDR_ERR_INVALID_ARG; }
    }
    if (in->N > 0 && (!in->shs || !in->scales || !in->rotations)) {
        set_error("views: needs shs + scales + rotations", hipSuccess);
        return GDR_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = launch_preprocess_bwd_views(V, s, in, geoms, radii, grad_recs, gout, st);
    if (e != hipSuccess) return hip_fail("preprocess_bwd_views", e);
    return debug_sync(&s[0], "preprocess_bwd_views", st);
}

int gdr_view_loss_forward(const float* color, const float* depth, const float* alpha, const float* target,
                          int32_t H, int32_t W, float w_depth, float w_alpha, float* loss, void* stream) {
    if (!color || !depth || !alpha || !target || !loss || H <= 0 || W <= 0) { set_error("view_loss_forward: bad argument", hipSuccess); return GDR_ERR_INVALID_ARG; }
    hipError_t e = launch_view_loss_fwd(color, depth, alpha, target, H * W, w_depth, w_alpha, loss, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail("view_loss_fwd", e);
    return GDR_OK;
}

int gdr_view_loss_backward(const float* color, const float* target, int32_t H, int32_t W, float w_depth,
                           float w_alpha, const float* g, float* dL_dcolor, float* dL_ddepth, float* dL_dalpha,
                           void* stream) {
    if (!color || !target || !dL_dcolor || !dL_ddepth || !dL_dalpha || H <= 0 || W <= 0) { set_error("view_loss_backward: bad argument", hipSuccess); return GDR_ERR_INVALID_ARG; }
    hipError_t e = launch_view_loss_bwd(color, target, H * W, w_depth, w_alpha, g, dL_dcolor, dL_ddepth, dL_dalpha, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail("view_loss_bwd", e);
    return GDR_OK;
}

int gdr_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return GDR_OK;
}

int gdr_profile_collect(double* ms_total, uint64_t* launches, int32_t n, int32_t reset) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof_pending) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(r.b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.a, r.b);
        if (e != hipSuccess) return hip_fail("profile_collect", e);
        g_prof_ms[r.id] += ms;
        g_prof_cnt[r.id] += 1;
        g_prof_pool.push_back(r.a);
        g_prof_pool.push_back(r.b);
    }
    g_prof_pending.clear();
    for (int k = 0; k < n && k < GDR_K_COUNT; ++k) {
        if (ms_total) ms_total[k] = g_prof_ms[k];
        if (launches) launches[k] = g_prof_cnt[k];
    }
    if (reset)
        for (int k = 0; k < GDR_K_COUNT; ++k) { g_prof_ms[k] = 0; g_prof_cnt[k] = 0; }
    return GDR_OK;
}

const char* gdr_kernel_name(int32_t id) {
    static const char* names[GDR_K_COUNT] = {"preprocess_fwd", "scan_block_sums", "duplicate_with_keys",
        "sort_hist", "sort_rowscan", "sort_scatter", "tile_ranges", "render_fwd", "render_bwd",
        "preprocess_bwd", "mark_visible", "tile_order", "tile_sort", "tile_sort_long", "view_loss", "surfel_maps", "knn",
        "topk_select", "render_fwd_deep", "tile_count", "tile_scan", "tile_scatter"};
    return (id >= 0 && id < GDR_K_COUNT) ? names[id] : "";
}
int gdr_kernel_count(void) { return GDR_K_COUNT; }

int gdr_mark_visible(int32_t N, const float* means3D, const float* viewmatrix,
                     const float* projmatrix, uint8_t* present, void* stream) {
    (void)projmatrix;
    if (N < 0 || (N > 0 && (!means3D || !viewmatrix || !present))) { set_error("mark_visible: bad argument", hipSuccess); return GDR_ERR_INVALID_ARG; }
    hipError_t e = launch_mark_visible(N, means3D, viewmatrix, present, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail("mark_visible", e);
    return GDR_OK;
}

}  // extern "C"

// =================================================================================
// 2DGS surfel path (include/gsr.h)
// =================================================================================
namespace gdr {
static size_t carve_surfel_geom(void* base, int64_t N, gdr_geom* g) {
    Carver c(base);
    gdr_geom t;
    const size_t n = (size_t)(N > 0 ? N : 1);
    t.depths = c.take<float>(n);
    t.rec = c.take<float>(GSR_REC_FLOATS * n);
    t.cov3D = nullptr;
    t.rect = c.take<int32_t>(4 * n);
    t.tiles_touched = c.take<uint32_t>(n);
    t.clamped = c.take<uint8_t>(n);
    t.block_sums = c.take<uint32_t>((n + GDR_BLOCK - 1) / GDR_BLOCK + 1);
    t.block_offs = c.take<uint32_t>((n + GDR_BLOCK - 1) / GDR_BLOCK + 1);
    t.num_rendered = c.take<uint32_t>(1);
    if (g) *g = t;
    return c.off;
}
static size_t carve_surfel_image(void* base, int H, int W, gdr_image* im) {
    Carver c(base);
    gdr_image t;
    const size_t tiles = (size_t)tile_grid_x(W) * tile_grid_y(H);
    const size_t P = (size_t)H * W;
    t.ranges = c.take<uint32_t>(2 * (tiles ? tiles : 1));
    t.n_contrib = c.take<uint32_t>(2 * (P ? P : 1));
    t.final_T = c.take<float>(3 * (P ? P : 1));
    t.tile_order = c.take<uint32_t>(tiles ? tiles : 1);
    t.seg_base = c.take<uint32_t>(tiles ? tiles : 1);
    if (im) *im = t;
    return c.off;
}
static int check_surfel(const gdr_settings* s, const gsr_inputs* in) {
    if (!s || !in) { set_error("NULL settings/inputs", hipSuccess); return GDR_ERR_INVALID_ARG; }
    if (in->N < 0 || s->image_height <= 0 || s->image_width <= 0) { set_error("negative N or empty image", hipSuccess); return GDR_ERR_INVALID_ARG; }
    if (in->N > GDR_MAX_GAUSSIANS) { set_error("N exceeds GDR_MAX_GAUSSIANS", hipSuccess); return GDR_ERR_UNSUPPORTED; }
    if (!s->bg || !s->viewmatrix || !s->projmatrix) { set_error("bg/viewmatrix/projmatrix must be device pointers", hipSuccess); return GDR_ERR_INVALID_ARG; }
    if (in->N > 0) {
        if (!in->means3D || !in->opacities) { set_error("means3D/opacities NULL", hipSuccess); return GDR_ERR_INVALID_ARG; }
        if ((in->shs != nullptr) == (in->colors_precomp != nullptr)) { set_error("provide exactly one of shs / colors_precomp", hipSuccess); return GDR_ERR_INVALID_ARG; }
        const bool sr = in->scales && in->rotations;
        if (sr == (in->transMat_precomp != nullptr) || ((in->scales != nullptr) != (in->rotations != nullptr))) {
            set_error("provide exactly one of (scales, rotations) / transMat_precomp", hipSuccess);
            return GDR_ERR_INVALID_ARG;
        }
        if (in->shs) {
            if (s->sh_degree < 0 || s->sh_degree > 3) { set_error("sh_degree must be 0..3", hipSuccess); return GDR_ERR_UNSUPPORTED; }
            if (in->M < (s->sh_degree + 1) * (s->sh_degree + 1)) { set_error("M < (sh_degree+1)^2", hipSuccess); return GDR_ERR_INVALID_ARG; }
            if (!s->campos) { set_error("campos NULL", hipSuccess); return GDR_ERR_INVALID_ARG; }
        }
    }
    return GDR_OK;
}
}  // namespace gdr

extern "C" {

size_t gsr_geom_bytes(int32_t N) { return carve_surfel_geom(nullptr, N, nullptr); }
size_t gsr_image_bytes(int32_t H, int32_t W) { return carve_surfel_image(nullptr, H, W, nullptr); }
int gsr_geom_carve(void* base, int32_t N, gdr_geom* out) {
    if (!base || !out || ((uintptr_t)base & 255u)) { set_error("geom base NULL/unaligned", hipSuccess); return GDR_ERR_INVALID_ARG; }
    carve_surfel_geom(base, N, out);
    return GDR_OK;
}
int gsr_image_carve(void* base, int32_t H, int32_t W, gdr_image* out) {
    if (!base || !out || ((uintptr_t)base & 255u)) { set_error("image base NULL/unaligned", hipSuccess); return GDR_ERR_INVALID_ARG; }
    carve_surfel_image(base, H, W, out);
    return GDR_OK;
}

int gsr_preprocess_forward(const gdr_settings* s, const gsr_inputs* in, const gdr_geom* geom, int32_t* radii,
                           uint32_t* num_rendered_host, void* stream) {
    int rc = check_surfel(s, in);
    if (rc) return rc;
    if (!geom || (in->N > 0 && !radii)) { set_error("geom/radii NULL", hipSuccess); return GDR_ERR_INVALID_ARG; }
    const int tiles = tile_grid_x(s->image_width) * tile_grid_y(s->image_height);
    if (key_bits(tiles) > 64) { set_error("image too large", hipSuccess); return GDR_ERR_UNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(geom->num_rendered, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return hip_fail("memset num_rendered", e);
    e = launch_surfel_preprocess_fwd(s, in, geom, radii, st);
    if (e != hipSuccess) return hip_fail("surfel_preprocess_fwd", e);
    if ((rc = debug_sync(s, "surfel_preprocess_fwd", st))) return rc;
    if (num_rendered_host) {
        e = hipMemcpyAsync(num_rendered_host, geom->num_rendered, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return hip_fail("memcpy num_rendered", e);
        e = hipStreamSynchronize(st);
        if (e != hipSuccess) return hip_fail("sync num_rendered", e);
    }
    return GDR_OK;
}

int gsr_composite_forward(const gdr_settings* s, const gdr_geom* geom, const gdr_binning* bin, const gdr_image* img,
                          const gsr_outputs* out, void* stream) {
    if (!s || !geom || !bin || !img || !out || !out->color || !out->allmap) {
        set_error("surfel composite_forward: NULL argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = launch_surfel_render_fwd(s, geom, bin, img, out, st);
    if (e != hipSuccess) return hip_fail("surfel_render_fwd", e);
    return debug_sync(s, "surfel_render_fwd", st);
}

int gsr_render_forward(const gdr_settings* s, const gsr_inputs* in, const gdr_geom* geom, gdr_binning* bin,
                       const gdr_image* img, uint64_t D, const gsr_outputs* out, void* stream) {
    int rc = check_surfel(s, in);
    if (rc) return rc;
    if (!geom || !bin || !img || !out || !out->color || !out->allmap || (in->N > 0 && !out->radii)) {
        set_error("surfel render_forward: NULL argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    if ((rc = binning_stage(s, in->N, geom, bin, img, D, out->radii, (hipStream_t)stream))) return rc;
    return gsr_composite_forward(s, geom, bin, img, out, stream);
}

int gsr_forward(const gdr_settings* s, const gsr_inputs* in, const gdr_geom* geom, gdr_binning* bin,
                const gdr_image* img, uint64_t D_cap, const gsr_outputs* out, uint32_t* num_rendered_host,
                void* stream) {
    if (!out || !num_rendered_host) { set_error("gsr_forward: NULL out/num_rendered_host", hipSuccess); return GDR_ERR_INVALID_ARG; }
    int rc = gsr_preprocess_forward(s, in, geom, out->radii, num_rendered_host, stream);
    if (rc) return rc;
    if ((uint64_t)*num_rendered_host > D_cap) { set_error("binning workspace too small for num_rendered", hipSuccess); return GDR_ERR_WORKSPACE; }
    return gsr_render_forward(s, in, geom, bin, img, *num_rendered_host, out, stream);
}

int gsr_backward(const gdr_settings* s, const gsr_inputs* in, const gdr_geom* geom, const gdr_binning* bin,
                 const gdr_image* img, uint64_t D, const int32_t* radii, const gsr_grad_inputs* gin,
                 const gsr_grad_outputs* gout, void* stream) {
    (void)D;
    int rc = check_surfel(s, in);
    if (rc) return rc;
    if (in->N == 0) return GDR_OK;  // no surfels: the (empty) gradient buffers may be NULL
    if (!geom || !bin || !img || !gin || !gout || !gin->dL_dcolor || !gout->dL_dmeans3D || !gout->dL_dmeans2D ||
        !gout->dL_dopacities || !gout->scratch || (in->N > 0 && !radii)) {
        set_error("surfel backward: NULL argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    if (in->shs && !gout->dL_dshs) { set_error("surfel backward: dL_dshs NULL", hipSuccess); return GDR_ERR_INVALID_ARG; }
    if (in->colors_precomp && !gout->dL_dcolors) { set_error("surfel backward: dL_dcolors NULL", hipSuccess); return GDR_ERR_INVALID_ARG; }
    if (in->transMat_precomp ? !gout->dL_dtransMat : (!gout->dL_dscales || !gout->dL_drotations)) {
        set_error("surfel backward: scale/rotation/transMat gradient buffers NULL", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t N = (size_t)in->N;
    if (N == 0) return GDR_OK;
    hipError_t e = hipMemsetAsync(gout->scratch, 0, N * GSR_GRAD_FLOATS * sizeof(float), st);
    if (e != hipSuccess) return hip_fail("memset gradient records", e);
    e = launch_surfel_render_bwd(s, geom, bin, img, gin, gout->scratch, st);
    if (e != hipSuccess) return hip_fail("surfel_render_bwd", e);
    if ((rc = debug_sync(s, "surfel_render_bwd", st))) return rc;
    e = launch_surfel_preprocess_bwd(s, in, geom, radii, gout, st);
    if (e != hipSuccess) return hip_fail("surfel_preprocess_bwd", e);
    return debug_sync(s, "surfel_preprocess_bwd", st);
}

static int check_surfel_views(int32_t V, const gdr_settings* s, const gsr_inputs* in, const gdr_geom* geoms) {
    if (V < 1 || V > GDR_MAX_VIEWS) { set_error("surfel views: V out of range", hipSuccess); return GDR_ERR_UNSUPPORTED; }
    if (!s || !in || !geoms) { set_error("surfel views: NULL argument", hipSuccess); return GDR_ERR_INVALID_ARG; }
    for (int v = 0; v < V; ++v) {
        int rc = check_surfel(&s[v], in);
        if (rc) return rc;
        if (s[v].image_width != s[0].image_width || s[v].image_height != s[0].image_height ||
            s[v].sh_degree != s[0].sh_degree || s[v].scale_modifier != s[0].scale_modifier) {
            set_error("surfel views: image size / sh_degree / scale_modifier must match", hipSuccess);
            return GDR_ERR_INVALID_ARG;
        }
    }
    if (in->N > 0 && (!in->shs || !in->scales || !in->rotations)) {
        set_error("surfel views: needs shs + scales + rotations", hipSuccess);
        return GDR_ERR_UNSUPPORTED;
    }
    return GDR_OK;
}

int gsr_preprocess_forward_views(int32_t V, const gdr_settings* s, const gsr_inputs* in, const gdr_geom* geoms,
                                 int32_t* const* radii, void* stream) {
    int rc = check_surfel_views(V, s, in, geoms);
    if (rc) return rc;
    if (!radii) { set_error("surfel views: radii NULL", hipSuccess); return GDR_ERR_INVALID_ARG; }
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    bool packed = true;  // the V counters in one array (geoms[v].num_rendered = base + v): one fill instead of V
    for (int v = 1; v < V; ++v) packed = packed && geoms[v].num_rendered == geoms[0].num_rendered + v;
    if (packed) e = hipMemsetAsync(geoms[0].num_rendered, 0, (size_t)V * sizeof(uint32_t), st);
    else
        for (int v = 0; v < V && e == hipSuccess; ++v) e = hipMemsetAsync(geoms[v].num_rendered, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return hip_fail("memset num_rendered", e);
    e = launch_surfel_preprocess_fwd_views(V, s, in, geoms, radii, st);
    if (e != hipSuccess) return hip_fail("surfel_preprocess_fwd_views", e);
    return debug_sync(&s[0], "surfel_preprocess_fwd_views", st);
}

int gsr_render_backward(const gdr_settings* s, int32_t N, const gdr_geom* geom, const gdr_binning* bin,
                        const gdr_image* img, const gsr_grad_inputs* gin, float* grad_rec, void* stream) {
    if (!s || !geom || !bin || !img || !gin || !gin->dL_dcolor || (N > 0 && !grad_rec) || !s->bg) {
        set_error("surfel render_backward: NULL argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    if (N <= 0) return GDR_OK;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = bin->grad_rec_cleared ? hipSuccess : hipMemsetAsync(grad_rec, 0, (size_t)N * GSR_GRAD_FLOATS * sizeof(float), st);
    if (e != hipSuccess) return hip_fail("memset gradient records", e);
    e = launch_surfel_render_bwd(s, geom, bin, img, gin, grad_rec, st);
    if (e != hipSuccess) return hip_fail("surfel_render_bwd", e);
    return debug_sync(s, "surfel_render_bwd", st);
}

int gsr_preprocess_backward_views(int32_t V, const gdr_settings* s, const gsr_inputs* in, const gdr_geom* geoms,
                                  const int32_t* const* radii, float* const* grad_recs, const gsr_grad_outputs* gout,
                                  void* stream) {
    if (V >= 1 && V <= GDR_MAX_VIEWS && s && in && in->N == 0) return GDR_OK;  // nothing to differentiate
    int rc = check_surfel_views(V, s, in, geoms);
    if (rc) return rc;
    if (!radii || !grad_recs || !gout || !gout->dL_dmeans3D || !gout->dL_dmeans2D || !gout->dL_dshs ||
        !gout->dL_dopacities || !gout->dL_dscales || !gout->dL_drotations) {
        set_error("surfel backward views: NULL argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    const int nb = (s[0].sh_degree + 1) * (s[0].sh_degree + 1);
    if ((3 * nb) % 4 == 0 && in->M != nb) {
        set_error("surfel backward views: M must equal (sh_degree+1)^2 at degrees 1 and 3", hipSuccess);
        return GDR_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = launch_surfel_preprocess_bwd_views(V, s, in, geoms, radii, grad_recs, gout, st);
    if (e != hipSuccess) return hip_fail("surfel_preprocess_bwd_views", e);
    return debug_sync(&s[0], "surfel_preprocess_bwd_views", st);
}

int gsr_maps_forward(const float* allmap, const float* rays, const float* viewmatrix, int32_t H, int32_t W,
                     float depth_ratio, float* depth, float* acc_map, float* rend_normal, float* depth_normal,
                     float* rend_dist, void* stream) {
    if (!allmap || !rays || !viewmatrix || !depth || !acc_map || !rend_normal || !depth_normal || !rend_dist || H <= 0 || W <= 0) {
        set_error("gsr_maps_forward: bad argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    hipError_t e = launch_surfel_maps_fwd(allmap, rays, viewmatrix, H, W, depth_ratio, depth, acc_map, rend_normal,
                                          depth_normal, rend_dist, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail("surfel_maps_fwd", e);
    return GDR_OK;
}

int gsr_maps_backward(const float* allmap, const float* rays, const float* viewmatrix, int32_t H, int32_t W,
                      float depth_ratio, const float* g_depth, const float* g_acc_map, const float* g_rend_normal,
                      const float* g_depth_normal, const float* g_rend_dist, float* scratch, float* dL_dallmap,
                      void* stream) {
    if (!allmap || !rays || !viewmatrix || !dL_dallmap || (g_depth_normal && !scratch) || H <= 0 || W <= 0) {
        set_error("gsr_maps_backward: bad argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    hipError_t e = launch_surfel_maps_bwd(allmap, rays, viewmatrix, H, W, depth_ratio, g_depth, g_acc_map, g_rend_normal,
                                          g_depth_normal, g_rend_dist, scratch, dL_dallmap, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail("surfel_maps_bwd", e);
    return GDR_OK;
}

int gsr_view_loss_forward(const float* color, const float* allmap, const float* rays, const float* viewmatrix,
                          const float* target, int32_t H, int32_t W, float depth_ratio, float w_dist, float w_normal,
                          float w_depth, float w_alpha, float* loss, void* stream) {
    if (!color || !allmap || !rays || !viewmatrix || !target || !loss || H <= 0 || W <= 0) {
        set_error("gsr_view_loss_forward: bad argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    hipError_t e = launch_surfel_loss_fwd(color, allmap, rays, viewmatrix, target, H, W, depth_ratio, w_dist, w_normal,
                                          w_depth, w_alpha, loss, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail("surfel_loss_fwd", e);
    return GDR_OK;
}

int gsr_view_loss_backward(const float* color, const float* allmap, const float* rays, const float* viewmatrix,
                           const float* target, int32_t H, int32_t W, float depth_ratio, float w_dist, float w_normal,
                           float w_depth, float w_alpha, const float* g, float* scratch, float* dL_dcolor,
                           float* dL_dallmap, void* stream) {
    if (!color || !allmap || !rays || !viewmatrix || !target || !scratch || !dL_dcolor || !dL_dallmap || H <= 0 || W <= 0) {
        set_error("gsr_view_loss_backward: bad argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    hipError_t e = launch_surfel_loss_bwd(color, allmap, rays, viewmatrix, target, H, W, depth_ratio, w_dist, w_normal,
                                          w_depth, w_alpha, g, scratch, dL_dcolor, dL_dallmap, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail("surfel_loss_bwd", e);
    return GDR_OK;
}

int gsr_knn_cells(const float* points, int32_t N, const float* bbox, int32_t G, int32_t* cell, void* stream) {
    if (N < 0 || G < 1 || G > 1024 || (N > 0 && (!points || !bbox || !cell))) { set_error("gsr_knn_cells: bad argument", hipSuccess); return GDR_ERR_INVALID_ARG; }
    hipError_t e = launch_knn_cells(points, N, bbox, G, cell, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail("knn_cells", e);
    return GDR_OK;
}

int gsr_knn_mean_dist2(const float* points_sorted, int32_t N, const float* bbox, int32_t G, const int32_t* cell_start,
                       float* out, void* stream) {
    if (N < 0 || G < 1 || G > 1024 || (N > 0 && (!points_sorted || !bbox || !cell_start || !out))) {
        set_error("gsr_knn_mean_dist2: bad argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    hipError_t e = launch_knn_mean_dist2(points_sorted, N, bbox, G, cell_start, out, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail("knn_mean_dist2", e);
    return GDR_OK;
}

}  // extern "C"
