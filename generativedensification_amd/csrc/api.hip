// api.hip — the C ABI of libgdr_hip.so (include/gdr.h): argument checking, workspace
// carving and stage sequencing.  Host code only; all kernels live in preprocess.hip,
// binning.hip and render.hip.  Nothing here allocates device memory; the only state kept between calls is the per-scene-shape
// history (duplicates per Gaussian and launch-size feedback behind gdr_view_plan_for, the K7 variant behind gdr_k7_tune_*) and
// small pools of pinned host words and events.
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <cmath>
#include <mutex>
#include <unordered_map>
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

// ---- which K7 for a scene shape: rows, or row pairs where they pay (render.hip: render_bwd_pairs_kernel) ---------------
// K7 is bound by the device's float-atomic line rate wherever the Gaussians span several 4x4 blocks; merging the two rows of
// an 8x4 area saves lines and costs iterations, and which side wins depends on the scene (C2 -16 %, object-like scenes and
// sub-pixel Gaussians +3-30 %).  Results are the same sums in another order.  Per (device, N bucket, duplicates-per-Gaussian
// class of the views, image size, views per launch, entry kind): ONE round of four consecutive launches is timed with events
// (rows / pairs / rows / pairs, the better try of each) after the key's first kK7First launches, and the faster variant (by
// > 3 %) serves that key from then on — the choice is made once and kept (round 4 repeated the round every 256 launches: the
// variant, and with it the order of the float sums, could change in the middle of a run; a scene that drifts — a densifying
// model — moves to another duplicates-per-Gaussian class, i.e. another key, instead).  gdr_k7_tune_override pins a variant
// (tests, A/B, bit-reproducible runs).
struct K7Tune {
    float us[2] = {0.f, 0.f};      // the running round's best time per variant (0 = not in yet)
    float us_round[2][2] = {{0.f, 0.f}, {0.f, 0.f}};   // what the first and the confirmation round measured (rows, pairs)
    uint32_t calls = 0;
    int chosen = 0, got = 0;
    int rounds_done = 0;           // 0: the first round is still to come / running, 1: decided once, 2: confirmed (final)
    bool decided = false;
    uint64_t last_use = 0;
    struct Pend { hipEvent_t a = nullptr, b = nullptr; int state = 0; } pend[4];   // state: 0 idle, 1 begun, 2 ended
};
static std::mutex g_k7_mu;
static std::unordered_map<uint64_t, K7Tune> g_k7;
static uint64_t g_k7_clock = 0;
static std::atomic<int> g_k7_override{-1};    // -1 measure and choose, 0 rows, 1 row pairs
// the round = four consecutive launches, rows / pairs / rows / pairs (the faster of two tries counts), after kK7First launches
// of the key — K7's duration drifts down over the first ten or so launches of a shape (C3: pairs 1332 -> 1202 us, rows 1335 ->
// 1277, profiles/r04_ab_k7_blocks.txt section 10), a round at launch 0 chose wrongly
// Round 6: a SECOND opinion.  One noisy timing (other streams busy, clocks still ramping) used to be the choice for the life
// of the process although a wrong pick costs 3 % (C4) to 15 % (C2); launches kK7Second..+3 of the key time both kernels once
// more and the first pick is overturned only if it loses that round by > 5 %.  After it the choice is final.
constexpr uint32_t kK7Round = 4, kK7First = 8, kK7Second = 64;

// duplicates per Gaussian of a view as a quarter-octave class (1..63; 0 = unknown: the state did not come from
// gdr_forward_view(s)) — what gdr_binning.k7_class carries from the forward to the K7 entry points
static int k7_class_of(uint64_t D, int N) {
    if (N <= 0 || D == 0) return 0;
    const double r = (double)D / (double)N;
    int c = 17 + (int)std::floor(4.0 * std::log2(r));
    return c < 1 ? 1 : (c > 63 ? 63 : c);
}
static int k7_class_views(int V, const gdr_binning* bins) {
    int c = 0;
    for (int v = 0; v < V; ++v) c = bins[v].k7_class > c ? bins[v].k7_class : c;
    return c & 63;
}
static uint64_t k7_key(int N, int H, int W, int V, int kind, int cls) {
    int dev = 0, nb = 0;
    (void)hipGetDevice(&dev);
    for (int64_t v = N; v > 0; v >>= 1) ++nb;
    return ((uint64_t)(dev & 0xFF) << 55) | ((uint64_t)(kind & 7) << 52) | ((uint64_t)(V & 0xF) << 48) |
           ((uint64_t)(nb & 0x3F) << 42) | ((uint64_t)(cls & 0x3F) << 36) | ((uint64_t)(H & 0x3FFFF) << 18) | (uint64_t)(W & 0x3FFFF);
}
static void k7_harvest(K7Tune& k) {   // (g_k7_mu held)
    for (uint32_t ph = 0; ph < kK7Round; ++ph) {
        K7Tune::Pend& p = k.pend[ph];
        if (p.state != 2 || hipEventQuery(p.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess && ms > 0.f) {
            float& u = k.us[ph & 1u];
            u = (u > 0.f && u < ms * 1e3f) ? u : ms * 1e3f;
            ++k.got;
        }
        p.state = 0;
    }
    if (k.rounds_done < 2 && k.got == (int)kK7Round && k.us[0] > 0.f && k.us[1] > 0.f) {
        if (k.rounds_done == 0) {
            k.chosen = k.us[1] < 0.97f * k.us[0] ? 1 : 0;
        } else {                                   // confirmation: switch only if the first pick clearly loses
            if (k.chosen == 0 && k.us[1] < 0.95f * k.us[0]) k.chosen = 1;
            else if (k.chosen == 1 && k.us[0] < 0.95f * k.us[1]) k.chosen = 0;
        }
        k.us_round[k.rounds_done][0] = k.us[0]; k.us_round[k.rounds_done][1] = k.us[1];
        k.us[0] = k.us[1] = 0.f;
        k.got = 0;
        k.decided = true;
        ++k.rounds_done;
    }
}
// brackets ONE K7 launch: sets the calling thread's variant, times the launch when it is this key's turn
struct K7Scope {
    K7Tune* t = nullptr;
    int slot = -1;
    hipStream_t st;
    K7Scope(int N, int H, int W, int V, int kind, int cls, hipStream_t stream) : st(stream) {
        int variant = g_k7_override.load();
        if (variant < 0) {
            std::lock_guard<std::mutex> lk(g_k7_mu);
            K7Tune& k = g_k7[k7_key(N, H, W, V, kind, cls)];
            k.last_use = ++g_k7_clock;
            k7_harvest(k);
            variant = k.chosen;
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            const bool capturing = hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
            if (capturing) (void)hipGetLastError();
            if (k.rounds_done < 2 && !capturing) {      // (a timing event cannot be queried on a stream under graph capture)
                const uint32_t start = k.rounds_done == 0 ? kK7First : kK7Second;
                const uint32_t phase = k.calls++ - start;                      // (wraps for the launches before the round)
                if (phase < kK7Round && k.pend[phase].state == 0) {
                    K7Tune::Pend& p = k.pend[phase];
                    if (phase == 0) { k.us[0] = k.us[1] = 0.f; k.got = 0; }
                    if (!p.a && (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess)) p.a = p.b = nullptr;
                    if (p.a && hipEventRecord(p.a, st) == hipSuccess) { p.state = 1; t = &k; slot = (int)phase; }
                    variant = (int)(phase & 1u);
                } else if (phase >= kK7Round && phase < 0x80000000u && k.got < (int)kK7Round) {
                    bool open = false;           // a try of the round was lost (an event call failed): start the round over
                    for (uint32_t ph = 0; ph < kK7Round; ++ph) open = open || k.pend[ph].state != 0;
                    if (!open) k.calls = start;
                }
            }
        }
        render_bwd_set_pairs(variant > 0 ? 1 : 0);
        surfel_render_bwd_set_pairs(variant > 0 ? 1 : 0);
    }
    ~K7Scope() {
        render_bwd_set_pairs(0);
        surfel_render_bwd_set_pairs(0);
        if (slot < 0) return;
        const bool ok = hipEventRecord(t->pend[slot].b, st) == hipSuccess;
        std::lock_guard<std::mutex> lk(g_k7_mu);
        t->pend[slot].state = ok ? 2 : 0;
    }
    K7Scope(const K7Scope&) = delete;
    K7Scope& operator=(const K7Scope&) = delete;
};

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
    t.hist_tiles = 0;
    t.k7_class = 0;
    t.reserved2 = 0;
    if (N > 0 && tiles > 0 && tiles <= GDR_BIN_MAX_TILES) {   // direct tile binning: (width rows x tiles) counts + a totals row
        int w = (N + 1023) / 1024;
        t.hist_width = w > GDR_BIN_MAX_WIDTH ? GDR_BIN_MAX_WIDTH : w;
        t.hist_tiles = (tiles + 63) / 64 * 64;
        t.tile_hist = c.take<uint32_t>((size_t)(t.hist_width + 1) * (size_t)t.hist_tiles);
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

// 2DGS surfel geometry / image state (include/gsr.h): the same structs, wider records
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
// stages (direct tile binning only; anything else runs whole under bit 0): 1 = count, scan, tile order; 2 = scatter; 4 = per-tile
// sort.  gdr_forward_views issues the stages of its views breadth-first — stage by stage over the views' streams — so that
// the last view's chain does not start a whole chain's worth of launches (9 x ~4 us of host time per view) after the first.
static int binning_stage_views(int V, const gdr_settings* s, int32_t N, const gdr_geom* geoms, gdr_binning* bins,
                               const gdr_image* imgs, const uint64_t* D, const int32_t* const* radii, hipStream_t st,
                               int stages = 7) {
    int rc;
    const int W = s->image_width, H = s->image_height;
    const int tiles = tile_grid_x(W) * tile_grid_y(H);
    hipError_t e;
    const bool global_sort = bins[0].global_sort != 0;
    bool direct = !global_sort && tiles <= GDR_BIN_MAX_TILES;
    uint64_t dmax = 0;
    for (int v = 0; v < V; ++v) {
        // (a count matrix carved for a smaller image cannot take this one: the radix partition produces the same lists)
        direct = direct && bins[v].tile_hist && bins[v].hist_width > 0 && bins[v].hist_width == bins[0].hist_width &&
                 bins[v].hist_tiles >= (tiles + 63) / 64 * 64;
        dmax = D[v] > dmax ? D[v] : dmax;
    }
    if (global_sort && (stages & 1)) {  // stable global sort: duplicates must be emitted in Gaussian order (once: the
        // breadth-first issue of gdr_forward_views calls this function with stages 1, 2 and 4, and the scan is in place)
        for (int v = 0; v < V; ++v) {
            e = launch_scan_block_sums(&geoms[v], N, st);
            if (e != hipSuccess) return hip_fail("scan_block_sums", e);
        }
    }
    BinViews vs;
    fill_bin_views(&vs, V, geoms, bins, imgs, D, radii);
    int sorted = 0;
    bool from_totals = false;
    if (!(stages & 1)) {      // a later stage of the direct path: the state stage 1 left
        if (!direct) return GDR_OK;
        from_totals = !(N == 0 || dmax == 0);
        for (int v = 0; v < V; ++v) vs.v[v].from_totals = from_totals ? 1 : 0;
    } else if (direct && (N == 0 || dmax == 0)) {   // nothing to bin: only the ranges are cleared (ranges_clear inside)
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
    if (!direct) stages = 7;
    if (stages & 1) {
        e = launch_tile_order_views(vs, V, tiles, st);  // (totals -> ranges first;) longest list first: launch order of tile_sort, K6, K7
        if (e != hipSuccess) return hip_fail("tile_order", e);
        if ((rc = debug_sync(s, "tile_order", st))) return rc;
    }
    if ((stages & 2) && from_totals) {
        e = launch_tile_scatter(vs, V, N, W, H, st);
        if (e != hipSuccess) return hip_fail("tile_scatter", e);
        if ((rc = debug_sync(s, "tile_scatter", st))) return rc;
    }
    if (!(stages & 4)) return GDR_OK;
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

// K6 of V views in one launch (round 4)
int gdr_composite_forward_views(int32_t V, const gdr_settings* s, const gdr_geom* geoms, const gdr_binning* bins,
                                const gdr_image* imgs, const gdr_outputs* outs, int32_t loss_mode,
                                const float* const* targets, float w_depth, float w_alpha, float go_scale, float* losses,
                                int32_t interleave, void* stream) {
    if (V < 1 || V > GDR_MAX_VIEWS) { set_error("views: V out of range", hipSuccess); return GDR_ERR_UNSUPPORTED; }
    if (!s || !geoms || !bins || !imgs || !outs || loss_mode < 0 || loss_mode > 2 || (loss_mode && (!targets || !losses))) {
        set_error("composite_forward_views: bad argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    for (int v = 0; v < V; ++v) {
        if (!s[v].bg || s[v].image_width != s[0].image_width || s[v].image_height != s[0].image_height || !outs[v].color ||
            (loss_mode != 2 && (!outs[v].depth || !outs[v].alpha)) || (loss_mode && !targets[v])) {
            set_error("composite_forward_views: NULL view buffer or image sizes differ", hipSuccess);
            return GDR_ERR_INVALID_ARG;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = launch_render_fwd_views(V, s, geoms, bins, imgs, outs, loss_mode, targets, w_depth, w_alpha, go_scale, losses,
                                           interleave, st);
    if (e != hipSuccess) return hip_fail("render_fwd_views", e);
    return debug_sync(&s[0], "render_fwd_views", st);
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
    { K7Scope k7(N, s->image_height, s->image_width, 1, 1, bin->k7_class, st);
      e = launch_render_bwd_loss(s, geom, bin, img, color, target, w_depth, w_alpha, g, grad_rec, st); }
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
    hipError_t e;
    { K7Scope k7(N, s->image_height, s->image_width, 1, 4, bin->k7_class, st);
      e = launch_render_bwd_mean2d_loss(s, geom, bin, img, color, target, g, dL_dmean2D, st); }
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
    { K7Scope k7((int)N, s->image_height, s->image_width, 1, 0, bin->k7_class, st);
      e = launch_render_bwd(s, geom, bin, img, gin, gout, st); }
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
    { K7Scope k7(N, s->image_height, s->image_width, 1, 0, bin->k7_class, st);
      e = launch_render_bwd(s, geom, bin, img, gin, &go, st); }
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
    hipError_t e;
    { K7Scope k7(N, s[0].image_height, s[0].image_width, V, 0, k7_class_views(V, bins), st);
      e = launch_render_bwd_views(V, s, geoms, bins, imgs, gins, grad_recs, interleave, st); }
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
    hipError_t e;
    { K7Scope k7(N, s[0].image_height, s[0].image_width, V, 1, k7_class_views(V, bins), st);
      e = launch_render_bwd_loss_views(V, s, geoms, bins, imgs, colors, targets, w_depth, w_alpha, g, grad_recs, interleave, st); }
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
    hipError_t e;
    { K7Scope k7(N, s[0].image_height, s[0].image_width, V, 2, k7_class_views(V, bins), st);
      e = launch_render_bwd_mean2d_views(V, s, geoms, bins, imgs, dL_dcolors, dL_dmean2D, interleave, st); }
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
    hipError_t e;
    { K7Scope k7(N, s->image_height, s->image_width, 1, 2, bin->k7_class, st);
      e = launch_render_bwd_mean2d(s, geom, bin, img, dL_dcolor, dL_dmean2D, st); }
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
// One forward call per view (round 4, v14): gdr_view_plan_for + gdr_forward_view (+ the surfel twins below).
// The reference's extension does its whole forward in ONE native call (`_C.rasterize_gaussians`, reached from
// /root/reference/lightning/renderer.py:250-259); until round 3 a call of the Python boundary was ~20 ctypes calls plus
// Python-side carving, capacity / launch-hint bookkeeping and a pooled read-back.  Here the library does all of that:
// carving of ONE caller allocation, K1, the duplicate count on its way to pinned host memory, binning + K6 sized by the
// device counter, the comparison with the capacity, and the per-shape history the next call is planned from.
// =================================================================================
namespace gdr {
namespace {

struct ShapeHist {
    double d_per_n = 0.0;          // decaying maximum of duplicates per Gaussian of one view of this shape (0: none yet)
    int64_t n_long = 0, n_medium = 0;   // decaying maxima of the tile sort's long / medium class sizes
    int deep_ttl = 0;              // calls until the deep forward launch may be dropped
    uint32_t* stats = nullptr;     // kStatViews x 4 pinned host words the binning stages of a call's views report into (kept for
                                   // the life of the process: a kernel in flight may still write them)
    bool reported = false;
};
std::mutex g_hist_mu;
std::unordered_map<uint64_t, ShapeHist> g_hist;
constexpr double kDSlack = 1.5;    // capacity = slack x the largest recent count of the shape (+ 4096)
constexpr int kStatViews = 64;     // report rows per shape (views beyond share the last row)

int bit_length(int64_t v) { int b = 0; while (v > 0) { ++b; v >>= 1; } return b; }

// key of the per-shape histories: device, power-of-two bucket of N (a densifying model renders a different N every step; what
// carries over between neighbouring N is the number of duplicates PER GAUSSIAN), image size, path (3DGS / surfel)
uint64_t shape_key(int N, int H, int W, int surfel) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return ((uint64_t)(dev & 0xFF) << 56) | ((uint64_t)(surfel & 1) << 55) | ((uint64_t)bit_length(N) << 48) |
           ((uint64_t)(H & 0xFFFFFF) << 24) | (uint64_t)(W & 0xFFFFFF);
}

struct PinnedWords { uint32_t* p; int dev; };
std::mutex g_pin_mu;
std::vector<PinnedWords> g_pin_pool;    // pinned buffers of kPinWords words for the count read-back (never freed: a handful)
constexpr int kPinWords = 256;          // >= GDR_MAX_NODE_VIEWS
uint32_t* pin_get(int dev) {
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        for (size_t k = 0; k < g_pin_pool.size(); ++k)
            if (g_pin_pool[k].dev == dev) { uint32_t* p = g_pin_pool[k].p; g_pin_pool[k] = g_pin_pool.back(); g_pin_pool.pop_back(); return p; }
    }
    uint32_t* p = nullptr;
    if (hipHostMalloc((void**)&p, kPinWords * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void pin_put(uint32_t* p, int dev) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pin_pool.push_back({p, dev});
}

int seg_len_policy(const gdr_view_opts* o, uint64_t d_est, int tiles, int64_t busy) {
    if (o && o->seg_len >= 0) return o->seg_len / GDR_BLOCK * GDR_BLOCK;
    // 512-entry segments pay on large images whose tiles are all busy with long lists (C4 +2.5 %, C5 +1.3 % over 256);
    // scenes with few busy tiles or short lists keep 256
    if (tiles >= 2000 && d_est >= (uint64_t)500 * (uint64_t)tiles && (busy < 0 || 2 * busy >= tiles)) return 512;
    return GDR_DEFAULT_SEG_LEN;
}

}  // namespace
}  // namespace gdr

extern "C" {

int gdr_view_plan_for(int32_t N, int32_t H, int32_t W, int32_t surfel, uint64_t exact_D, const gdr_view_opts* opts,
                      gdr_view_plan* plan) {
    if (!plan || N < 0 || H <= 0 || W <= 0) { set_error("view_plan_for: bad argument", hipSuccess); return GDR_ERR_INVALID_ARG; }
    const int tiles = tile_grid_x(W) * tile_grid_y(H);
    uint64_t cap = exact_D;
    int64_t busy = -1;
    int deferred = 0;
    if (!exact_D && N > 0) {
        std::lock_guard<std::mutex> lk(g_hist_mu);
        auto it = g_hist.find(shape_key(N, H, W, surfel));
        if (it != g_hist.end() && it->second.d_per_n > 0.0) {
            cap = (uint64_t)(it->second.d_per_n * (double)N * kDSlack) + 4096;
            deferred = 1;
            if (it->second.reported && it->second.stats && it->second.stats[0] != 0xFFFFFFFFu) busy = it->second.stats[3];
        }
    }
    if (cap > GDR_MAX_RENDERED) cap = GDR_MAX_RENDERED;
    plan->capacity = cap;
    plan->deferred = deferred;
    plan->have_binning = (exact_D || deferred || N == 0) ? 1 : 0;
    plan->seg_len = seg_len_policy(opts, deferred ? (uint64_t)((double)cap / kDSlack) : cap, tiles, busy);
    const bool direct = !(opts && (opts->radix_partition || opts->global_sort));
    size_t geom = surfel ? carve_surfel_geom(nullptr, N, nullptr) : carve_geom(nullptr, N, nullptr);
    size_t img = surfel ? carve_surfel_image(nullptr, H, W, nullptr) : carve_image(nullptr, H, W, nullptr);
    size_t bin = plan->have_binning ? carve_binning(nullptr, cap, nullptr, plan->seg_len, direct ? N : 0, direct ? tiles : 0) : 0;
    plan->bytes = (uint64_t)(geom + img + bin);
    return GDR_OK;
}

}  // extern "C"

namespace gdr {
namespace {

// launch-size feedback of a scene shape: the report words of its previous call(s) -> this call's hints (decaying maxima over
// the views and the recent calls + 25 %, never below 16 / 32 workgroups: a scene that suddenly has a hundred long lists costs a
// few rounds on a small grid, not one workgroup sorting them all); stats = the kStatViews x 4 pinned words this call reports into
struct ShapeHints { uint32_t* stats; int hint_long, hint_medium, hint_no_deep; };
ShapeHints shape_hints(uint64_t key, int N, const gdr_view_opts* opts) {
    ShapeHints out{nullptr, 0, 0, 0};
    if ((opts && opts->no_hints) || N <= 0) return out;
    std::lock_guard<std::mutex> lk(g_hist_mu);
    if (g_hist.size() >= 4096 && !g_hist.count(key)) return out;
    ShapeHist& h = g_hist[key];
    if (!h.stats) {
        if (hipHostMalloc((void**)&h.stats, kStatViews * 4 * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess)
            for (int k = 0; k < kStatViews * 4; ++k) h.stats[k] = 0xFFFFFFFFu;
        else { h.stats = nullptr; (void)hipGetLastError(); }
    }
    out.stats = h.stats;
    if (!h.stats) return out;
    int64_t n_long = -1, n_medium = 0;
    bool deep = false;
    for (int r = 0; r < kStatViews; ++r) {
        const uint32_t* w = h.stats + 4 * r;
        if (w[0] == 0xFFFFFFFFu) continue;
        n_long = std::max<int64_t>(n_long, w[0]); n_medium = std::max<int64_t>(n_medium, w[1]); deep = deep || w[2] != 0;
    }
    if (n_long < 0) return out;       // nothing reported yet
    h.reported = true;
    h.n_long = std::max<int64_t>(n_long, h.n_long * 9 / 10);
    h.n_medium = std::max<int64_t>(n_medium, h.n_medium * 9 / 10);
    h.deep_ttl = deep ? 8 : std::max(0, h.deep_ttl - 1);
    out.hint_long = h.n_long == 0 ? -1 : (int)std::max<int64_t>(16, h.n_long + h.n_long / 4 + 1);
    out.hint_medium = (int)std::max<int64_t>(32, h.n_medium + h.n_medium / 4 + 1);
    out.hint_no_deep = h.deep_ttl == 0 ? 1 : 0;
    return out;
}

void record_duplicates(uint64_t key, int N, uint64_t d_max) {   // the history the next call of this shape is planned from
    if (N <= 0) return;
    std::lock_guard<std::mutex> lk(g_hist_mu);
    if (g_hist.size() >= 4096 && !g_hist.count(key)) return;
    ShapeHist& h = g_hist[key];
    h.d_per_n = std::max((double)d_max / (double)std::max(N, 1), h.d_per_n * 0.97);
    if (h.d_per_n <= 0.0) h.d_per_n = 1e-9;    // "seen": a shape without duplicates still gets device-sized calls
}

// pooled events (no timing): cross-stream ordering inside gdr_forward_views.  An event goes back to the pool as soon as the
// waits on it are enqueued (a wait refers to the record that preceded it; a later re-record does not disturb it).
std::mutex g_ev_mu;
std::vector<hipEvent_t> g_ev_pool;
hipEvent_t event_get() {
    {
        std::lock_guard<std::mutex> lk(g_ev_mu);
        if (!g_ev_pool.empty()) { hipEvent_t e = g_ev_pool.back(); g_ev_pool.pop_back(); return e; }
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    return e;
}
void event_put(hipEvent_t e) {
    if (!e) return;
    std::lock_guard<std::mutex> lk(g_ev_mu);
    g_ev_pool.push_back(e);
}

// the forward of one view, shared by the 3DGS and the surfel boundary: K1 and K6 come in as callables
template <class K1, class K6>
int forward_view_impl(const gdr_settings* s, int N, int surfel, const gdr_view_plan* plan, void* ws, const gdr_view_opts* opts,
                      const gdr_same_as* same, int32_t* radii, gdr_view_state* out_st, hipStream_t st, K1 run_k1, K6 run_k6) {
    const int W = s->image_width, H = s->image_height;
    const int tiles = tile_grid_x(W) * tile_grid_y(H);
    if (!ws || ((uintptr_t)ws & 255u)) { set_error("forward_view: workspace NULL / not 256-byte aligned", hipSuccess); return GDR_ERR_INVALID_ARG; }
    if (key_bits(tiles) > 64) { set_error("image too large", hipSuccess); return GDR_ERR_UNSUPPORTED; }
    gdr_view_state v;
    memset(&v, 0, sizeof(v));
    char* base = (char*)ws;
    size_t off = surfel ? carve_surfel_geom(base, N, &v.geom) : carve_geom(base, N, &v.geom);
    off += surfel ? carve_surfel_image(base + off, H, W, &v.img) : carve_image(base + off, H, W, &v.img);
    const bool direct = !(opts && (opts->radix_partition || opts->global_sort));
    if (plan->have_binning) {
        const size_t b = carve_binning(base + off, plan->capacity, &v.bin, plan->seg_len, direct ? N : 0, direct ? tiles : 0);
        if ((uint64_t)(off + b) > plan->bytes) { set_error("forward_view: plan does not match its byte count", hipSuccess); return GDR_ERR_WORKSPACE; }
    }
    // launch-size feedback of the shape (results never depend on it, include/gdr.h gdr_binning.stats_out / hint_*)
    const uint64_t key = shape_key(N, H, W, surfel);
    const ShapeHints hn = shape_hints(key, N, opts);
    uint32_t* stats = hn.stats;
    const int hint_long = hn.hint_long, hint_medium = hn.hint_medium, hint_no_deep = hn.hint_no_deep;
    auto apply_opts = [&](gdr_binning& b) {
        b.global_sort = (opts && opts->global_sort) ? 1 : 0;
        if (opts && opts->deep_max_busy >= 0) b.deep_max_busy = opts->deep_max_busy;
        if (opts && opts->deep_min_mean >= 0) b.deep_min_mean = opts->deep_min_mean;
        b.hint_long = hint_long; b.hint_medium = hint_medium; b.hint_no_deep = hint_no_deep;
        b.stats_out = stats;
    };
    hipError_t e = hipMemsetAsync(v.geom.num_rendered, 0, 2 * sizeof(uint32_t), st);   // [count, same_as verdict]
    if (e != hipSuccess) return hip_fail("memset num_rendered", e);
    if (same && same->n > 0) {
        if (same->n > GDR_DIFFER_MAX) { set_error("forward_view: more than GDR_SAME_AS_MAX same_as pairs", hipSuccess); return GDR_ERR_INVALID_ARG; }
        e = launch_words_differ_multi(same->n, same->a, same->b, same->n_bytes, v.geom.num_rendered + 1, st);
        if (e != hipSuccess) return hip_fail("words_differ_multi", e);
    }
    int rc = run_k1(v.geom);
    if (rc) return rc;
    // the count (and the verdict word behind it) on its way to pinned host memory
    int dev = 0;
    (void)hipGetDevice(&dev);
    HostCopyTicket* ticket = nullptr;
    uint32_t* pin = pin_get(dev);
    if (!pin) return hip_fail("hipHostMalloc", hipErrorOutOfMemory);
    auto release_pin = [&]() { pin_put(pin, dev); };
    rc = gdr_host_copy_begin(pin, v.geom.num_rendered, 2 * sizeof(uint32_t), (void*)st, (void**)&ticket);
    if (rc) { release_pin(); return rc; }
    auto finish = [&](int code) { *out_st = v; return code; };
    uint64_t D = 0;
    if (plan->have_binning && plan->deferred) {   // device-sized call: binning + K6 are enqueued behind K1 without waiting for it
        apply_opts(v.bin);
        v.bin.d_dev = v.geom.num_rendered;
        rc = binning_stage(s, N, &v.geom, &v.bin, &v.img, plan->capacity, radii, st);
        if (!rc) rc = run_k6(v.geom, v.bin, v.img);
        const int wrc = gdr_host_copy_wait(ticket);
        D = pin[0]; v.differ = pin[1];
        release_pin();
        if (rc) return rc;
        if (wrc) return wrc;
    } else {
        rc = gdr_host_copy_wait(ticket);
        D = pin[0]; v.differ = pin[1];
        release_pin();
        if (rc) return rc;
        if (plan->have_binning && D <= plan->capacity) {
            apply_opts(v.bin);
            v.bin.d_dev = nullptr;
            rc = binning_stage(s, N, &v.geom, &v.bin, &v.img, D, radii, st);
            if (!rc) rc = run_k6(v.geom, v.bin, v.img);
            if (rc) return rc;
        }
    }
    v.D = D;
    v.bin.k7_class = k7_class_of(D, N);
    record_duplicates(key, N, D);
    if (!plan->have_binning || D > plan->capacity) {
        set_error("forward_view: binning workspace too small for num_rendered (plan again with exact_D = state.D)", hipSuccess);
        return finish(GDR_ERR_WORKSPACE);
    }
    return finish(GDR_OK);
}

}  // namespace
}  // namespace gdr

extern "C" {

int gdr_forward_view(const gdr_settings* s, const gdr_inputs* in, const gdr_view_plan* plan, void* workspace,
                     const gdr_view_opts* opts, const gdr_same_as* same, const gdr_outputs* out, gdr_view_state* state,
                     void* stream) {
    int rc = check_common(s, in);
    if (rc) return rc;
    if (!plan || !out || !state || !out->color || !out->depth || !out->alpha || (in->N > 0 && !out->radii)) {
        set_error("forward_view: NULL argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    return forward_view_impl(
        s, in->N, 0, plan, workspace, opts, same, out->radii, state, st,
        [&](const gdr_geom& g) -> int {
            hipError_t e = launch_preprocess_fwd(s, in, &g, out->radii, st);
            if (e != hipSuccess) return hip_fail("preprocess_fwd", e);
            return debug_sync(s, "preprocess_fwd", st);
        },
        [&](const gdr_geom& g, const gdr_binning& b, const gdr_image& im) -> int {
            return gdr_composite_forward(s, &g, &b, &im, out, stream);
        });
}

int gdr_view_reuse_probe(const gdr_settings* s, int32_t n, const gdr_settings* candidates, const gdr_same_as* same,
                         uint32_t* scratch, int32_t* match, uint32_t* differ, void* stream) {
    static_assert(GDR_DIFFER_MAX == GDR_SAME_AS_MAX, "gdr_same_as arrays");
    static_assert(GDR_REUSE_MAX + 1 <= kPinWords, "reuse probe read-back buffer");
    if (!s || n < 0 || n > GDR_REUSE_MAX || (n && !candidates) || !scratch || !match || !differ || !s->bg || !s->viewmatrix ||
        !s->projmatrix || (same && (same->n < 0 || same->n > GDR_SAME_AS_MAX))) {
        set_error("view_reuse_probe: bad argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    *match = -1;
    *differ = 0;
    // the host half of the 12 fields; a candidate without a campos tensor only matches a call without one
    uint64_t eligible = 0;
    gdr_settings cur = *s;
    gdr_settings cand[GDR_REUSE_MAX];
    if (!cur.campos) cur.campos = cur.bg;
    for (int c = 0; c < n; ++c) {
        const gdr_settings& k = candidates[c];
        cand[c] = k;
        if (!k.bg || !k.viewmatrix || !k.projmatrix || (k.campos == nullptr) != (s->campos == nullptr)) continue;
        if (!cand[c].campos) cand[c].campos = cand[c].bg;
        if (k.image_height == s->image_height && k.image_width == s->image_width && k.tanfovx == s->tanfovx &&
            k.tanfovy == s->tanfovy && k.scale_modifier == s->scale_modifier && k.sh_degree == s->sh_degree &&
            (k.prefiltered != 0) == (s->prefiltered != 0) && (k.debug != 0) == (s->debug != 0))
            eligible |= 1ull << c;
    }
    const bool pairs = same && same->n > 0;
    if (!eligible && !pairs) return GDR_OK;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(scratch, 0, (GDR_REUSE_MAX + 1) * sizeof(uint32_t), st);
    if (e != hipSuccess) return hip_fail("view_reuse_probe: memset", e);
    if (pairs) {
        e = launch_words_differ_multi(same->n, same->a, same->b, same->n_bytes, scratch + GDR_REUSE_MAX, st);
        if (e != hipSuccess) return hip_fail("words_differ_multi", e);
    }
    if (eligible) {
        e = launch_settings_match(&cur, n, cand, eligible, scratch, st);
        if (e != hipSuccess) return hip_fail("settings_match", e);
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    uint32_t* pin = pin_get(dev);
    if (!pin) return hip_fail("hipHostMalloc", hipErrorOutOfMemory);
    HostCopyTicket* ticket = nullptr;
    int rc = gdr_host_copy_begin(pin, scratch, (GDR_REUSE_MAX + 1) * sizeof(uint32_t), stream, (void**)&ticket);
    if (!rc) rc = gdr_host_copy_wait(ticket);
    if (!rc) {
        for (int c = 0; c < n; ++c)
            if (((eligible >> c) & 1ull) && pin[c]) { *match = c; break; }
        *differ = pin[GDR_REUSE_MAX];
    }
    pin_put(pin, dev);
    return rc;
}

double gdr_view_history_get(int32_t N, int32_t H, int32_t W, int32_t surfel) {
    std::lock_guard<std::mutex> lk(g_hist_mu);
    auto it = g_hist.find(shape_key(N, H, W, surfel));
    return it == g_hist.end() ? 0.0 : it->second.d_per_n;
}

void gdr_view_history_set(int32_t N, int32_t H, int32_t W, int32_t surfel, double d_per_n) {
    std::lock_guard<std::mutex> lk(g_hist_mu);
    g_hist[shape_key(N, H, W, surfel)].d_per_n = d_per_n > 0.0 ? d_per_n : 0.0;
}

// ---- all views of one Gaussian set in ONE native call (the forward of the multi-view node) -------------------------------
int gdr_views_plan_for(int32_t V, int32_t N, int32_t H, int32_t W, uint64_t exact_D, const gdr_view_opts* opts,
                       gdr_views_plan* plan) {
    if (!plan || V < 1 || V > GDR_MAX_NODE_VIEWS) { set_error("views_plan_for: bad argument", hipSuccess); return GDR_ERR_INVALID_ARG; }
    gdr_view_plan one;
    int rc = gdr_view_plan_for(N, H, W, 0, exact_D, opts, &one);
    if (rc) return rc;
    plan->view = one;
    plan->V = V;
    plan->bytes_view = (one.bytes + 255) / 256 * 256;
    plan->bytes_shared = 256 + (uint64_t)((V * sizeof(uint32_t) + 255) / 256 * 256);     // the V packed duplicate counters
    plan->bytes = (uint64_t)V * plan->bytes_view + plan->bytes_shared;
    return GDR_OK;
}

int gdr_forward_views(int32_t V, const gdr_settings* s, const gdr_inputs* in, const gdr_views_plan* plan, void* workspace,
                      const gdr_view_opts* opts, const gdr_outputs* outs, int32_t loss_mode, const float* const* targets,
                      float w_depth, float w_alpha, float go_scale, float* losses, void* const* streams, int32_t n_streams,
                      gdr_view_state* states) {
    if (!plan || V < 1 || V != plan->V || V > GDR_MAX_NODE_VIEWS || !s || !in || !outs || !states || !streams || n_streams < 1 ||
        loss_mode < 0 || loss_mode > 2 || (loss_mode && (!targets || !losses))) {
        set_error("forward_views: bad argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    if (!workspace || ((uintptr_t)workspace & 255u)) { set_error("forward_views: workspace NULL / not 256-byte aligned", hipSuccess); return GDR_ERR_INVALID_ARG; }
    const int N = in->N, W = s[0].image_width, H = s[0].image_height;
    const int tiles = tile_grid_x(W) * tile_grid_y(H);
    for (int v = 0; v < V; ++v) {
        int rc = check_common(&s[v], in);
        if (rc) return rc;
        if (s[v].image_width != W || s[v].image_height != H || s[v].sh_degree != s[0].sh_degree || s[v].scale_modifier != s[0].scale_modifier) {
            set_error("forward_views: image size / sh_degree / scale_modifier must match", hipSuccess);
            return GDR_ERR_INVALID_ARG;
        }
        if (!outs[v].color || (loss_mode != 2 && (!outs[v].depth || !outs[v].alpha)) || (N > 0 && !outs[v].radii) || (loss_mode && !targets[v])) {
            set_error("forward_views: NULL view buffer", hipSuccess);
            return GDR_ERR_INVALID_ARG;
        }
    }
    if (N > 0 && (!in->shs || !in->scales || !in->rotations)) { set_error("views: needs shs + scales + rotations", hipSuccess); return GDR_ERR_UNSUPPORTED; }
    if (key_bits(tiles) > 64) { set_error("image too large", hipSuccess); return GDR_ERR_UNSUPPORTED; }
    const gdr_view_plan& pv = plan->view;
    const bool direct = !(opts && (opts->radix_partition || opts->global_sort));
    char* base = (char*)workspace;
    uint32_t* counters = (uint32_t*)(base + (size_t)V * plan->bytes_view + 256);
    std::vector<gdr_geom> geoms((size_t)V);
    for (int v = 0; v < V; ++v) {
        gdr_view_state& st = states[v];
        memset(&st, 0, sizeof(st));
        char* b = base + (size_t)v * plan->bytes_view;
        size_t off = carve_geom(b, N, &st.geom);
        off += carve_image(b + off, H, W, &st.img);
        if (pv.have_binning) carve_binning(b + off, pv.capacity, &st.bin, pv.seg_len, direct ? N : 0, direct ? tiles : 0);
        st.geom.cov3D = states[0].geom.cov3D;          // view-independent: one copy
        st.geom.num_rendered = counters + v;
        geoms[(size_t)v] = st.geom;
    }
    hipStream_t s0 = (hipStream_t)streams[0];
    hipError_t e = hipMemsetAsync(counters, 0, (size_t)V * sizeof(uint32_t), s0);
    if (e != hipSuccess) return hip_fail("memset num_rendered", e);
    for (int lo = 0; lo < V; lo += GDR_MAX_VIEWS) {     // K1: inputs read once per <= 8 views
        const int n = V - lo < GDR_MAX_VIEWS ? V - lo : GDR_MAX_VIEWS;
        int32_t* rad[GDR_MAX_VIEWS];
        for (int k = 0; k < n; ++k) rad[k] = outs[lo + k].radii;
        e = launch_preprocess_fwd_views(n, s + lo, in, geoms.data() + lo, rad, s0);
        if (e != hipSuccess) return hip_fail("preprocess_fwd_views", e);
    }
    int rc = debug_sync(&s[0], "preprocess_fwd_views", s0);
    if (rc) return rc;
    const int ns = n_streams < V ? n_streams : V;
    if (ns > 1) {       // the side streams start behind K1
        hipEvent_t ready = event_get();
        e = hipEventRecord(ready, s0);
        for (int k = 1; k < ns && e == hipSuccess; ++k) e = hipStreamWaitEvent((hipStream_t)streams[k], ready, 0);
        event_put(ready);
        if (e != hipSuccess) return hip_fail("forward_views: stream fork", e);
    }
    // the V counts on their way to pinned host memory, in front of the LAST chain (the caller's stream goes straight on)
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::vector<uint32_t> d_host((size_t)V);
    uint32_t* pin = pin_get(dev);
    if (!pin) return hip_fail("hipHostMalloc", hipErrorOutOfMemory);
    struct PinBack { uint32_t* p; int dev; ~PinBack() { pin_put(p, dev); } } pin_back{pin, dev};
    static_assert(GDR_MAX_NODE_VIEWS <= kPinWords, "count read-back buffer");
    HostCopyTicket* ticket = nullptr;
    rc = gdr_host_copy_begin(pin, counters, (uint64_t)V * sizeof(uint32_t), streams[ns - 1], (void**)&ticket);
    if (rc) return rc;
    const uint64_t key = shape_key(N, H, W, 0);
    const ShapeHints hn = shape_hints(key, N, opts);
    // stage: 0 = the whole chain of a view; 1 / 2 / 4 = one binning stage (binning_stage_views); 8 = K6
    auto chain = [&](int v, hipStream_t st, uint64_t D, bool deferred, int stage) -> int {
        gdr_view_state& vs = states[v];
        if (stage == 0 || stage == 1) {
            vs.bin.global_sort = (opts && opts->global_sort) ? 1 : 0;
            if (opts && opts->deep_max_busy >= 0) vs.bin.deep_max_busy = opts->deep_max_busy;
            if (opts && opts->deep_min_mean >= 0) vs.bin.deep_min_mean = opts->deep_min_mean;
            vs.bin.hint_long = hn.hint_long; vs.bin.hint_medium = hn.hint_medium; vs.bin.hint_no_deep = hn.hint_no_deep;
            vs.bin.stats_out = hn.stats ? hn.stats + 4 * (v < kStatViews ? v : kStatViews - 1) : nullptr;
            vs.bin.d_dev = deferred ? vs.geom.num_rendered : nullptr;
        }
        if (stage != 8) {
            const uint64_t Dv = D;
            const int32_t* rad = outs[v].radii;
            int r = binning_stage_views(1, &s[v], N, &vs.geom, &vs.bin, &vs.img, &Dv, &rad, st, stage ? stage : 7);
            if (r || stage) return r;
        }
        if (loss_mode == 1) return gdr_composite_forward_loss(&s[v], &vs.geom, &vs.bin, &vs.img, &outs[v], targets[v], w_depth, w_alpha, losses + v, (void*)st);
        if (loss_mode == 2) return gdr_composite_forward_lossgrad(&s[v], &vs.geom, &vs.bin, &vs.img, targets[v], go_scale, losses + v, outs[v].color, (void*)st);
        return gdr_composite_forward(&s[v], &vs.geom, &vs.bin, &vs.img, &outs[v], (void*)st);
    };
    // the views' chains, issued breadth-first over their streams (stage by stage) when there is more than one stream: a chain
    // is 9 launches = ~40 us of host time, and issued view after view the last of 8 views starts 300 us after the first
    // (same-box A/B, profiles/r04_ab_k7_blocks.txt section 15: object-like C3 scenes +5 %, the reference's per-sample sequence on
    // them +3 %, uniform scenes +-0)
    auto issue = [&](bool deferred) -> int {
        int r = 0;
        if (ns <= 1) {
            for (int v = 0; v < V && !r; ++v) r = chain(v, (hipStream_t)streams[v % ns], deferred ? pv.capacity : d_host[(size_t)v], deferred, 0);
            return r;
        }
        for (int stage = 1; stage <= 8 && !r; stage <<= 1)
            for (int v = 0; v < V && !r; ++v)
                r = chain(v, (hipStream_t)streams[v % ns], deferred ? pv.capacity : d_host[(size_t)v], deferred, stage);
        return r;
    };
    auto join = [&]() -> hipError_t {      // the caller's stream continues only after every view is rendered
        hipError_t er = hipSuccess;
        for (int k = 1; k < ns && er == hipSuccess; ++k) {
            hipEvent_t done = event_get();
            er = hipEventRecord(done, (hipStream_t)streams[k]);
            if (er == hipSuccess) er = hipStreamWaitEvent(s0, done, 0);
            event_put(done);
        }
        return er;
    };
    bool fits = pv.have_binning != 0;
    if (pv.have_binning && pv.deferred) {     // device-sized: every chain is enqueued without waiting for K1
        rc = issue(true);
        e = join();
        const int wrc = gdr_host_copy_wait(ticket);
        if (rc) return rc;
        if (e != hipSuccess) return hip_fail("forward_views: stream join", e);
        if (wrc) return wrc;
        for (int v = 0; v < V; ++v) { d_host[(size_t)v] = pin[v]; fits = fits && pin[v] <= pv.capacity; }
    } else {
        rc = gdr_host_copy_wait(ticket);
        if (rc) return rc;
        for (int v = 0; v < V; ++v) { d_host[(size_t)v] = pin[v]; fits = fits && pin[v] <= pv.capacity; }
        if (fits) {
            rc = issue(false);
            e = join();
            if (rc) return rc;
            if (e != hipSuccess) return hip_fail("forward_views: stream join", e);
        } else if (ns > 1) {
            e = join();
            if (e != hipSuccess) return hip_fail("forward_views: stream join", e);
        }
    }
    uint64_t d_max = 0;
    for (int v = 0; v < V; ++v) {
        states[v].D = d_host[(size_t)v];
        states[v].bin.k7_class = k7_class_of(d_host[(size_t)v], N);
        d_max = d_max > d_host[(size_t)v] ? d_max : d_host[(size_t)v];
    }
    record_duplicates(key, N, d_max);
    if (!fits) {
        set_error("forward_views: binning workspace too small for num_rendered (plan again with exact_D = the largest state.D)", hipSuccess);
        return GDR_ERR_WORKSPACE;
    }
    return GDR_OK;
}

int gdr_view_history_report(int32_t N, int32_t H, int32_t W, int32_t surfel, int32_t row, uint32_t* words, int32_t set) {
    if (row < 0 || row >= kStatViews || !words) { set_error("view_history_report: bad argument", hipSuccess); return GDR_ERR_INVALID_ARG; }
    std::lock_guard<std::mutex> lk(g_hist_mu);
    auto it = g_hist.find(shape_key(N, H, W, surfel));
    if (it == g_hist.end() || !it->second.stats) { for (int k = 0; k < 4 && !set; ++k) words[k] = 0xFFFFFFFFu; return set ? GDR_ERR_INVALID_ARG : GDR_OK; }
    for (int k = 0; k < 4; ++k) {
        if (set) it->second.stats[4 * row + k] = words[k];
        else words[k] = it->second.stats[4 * row + k];
    }
    if (set) { it->second.n_long = it->second.n_medium = 0; it->second.deep_ttl = 0; }    // (the decaying maxima restart from the words)
    return GDR_OK;
}

void gdr_k7_tune_override(int32_t mode) { g_k7_override.store(mode < 0 ? -1 : (mode ? 1 : 0)); }

int gdr_k7_tune_get(int32_t N, int32_t H, int32_t W, int32_t V, int32_t kind, int32_t* chosen, float* us_rows, float* us_pairs) {
    std::lock_guard<std::mutex> lk(g_k7_mu);
    // the most recently used key of this (N bucket, image size, views per launch, kind), whatever its duplicates class
    const uint64_t probe = k7_key(N, H, W, V, kind, 0), mask = ~((uint64_t)0x3F << 36);
    K7Tune* best = nullptr;
    for (auto& kv : g_k7)
        if ((kv.first & mask) == (probe & mask) && kv.second.last_use && (!best || kv.second.last_use > best->last_use)) best = &kv.second;
    if (!best) { set_error("k7_tune_get: no launch of this shape yet", hipSuccess); return GDR_ERR_INVALID_ARG; }
    k7_harvest(*best);
    if (chosen) *chosen = best->decided ? best->chosen : -1;
    if (us_rows) *us_rows = best->rounds_done ? best->us_round[0][0] : best->us[0];
    if (us_pairs) *us_pairs = best->rounds_done ? best->us_round[0][1] : best->us[1];
    return GDR_OK;
}

static K7Tune* k7_latest(int32_t N, int32_t H, int32_t W, int32_t V, int32_t kind) {   // (g_k7_mu held)
    const uint64_t probe = k7_key(N, H, W, V, kind, 0), mask = ~((uint64_t)0x3F << 36);
    K7Tune* best = nullptr;
    for (auto& kv : g_k7)
        if ((kv.first & mask) == (probe & mask) && kv.second.last_use && (!best || kv.second.last_use > best->last_use)) best = &kv.second;
    return best;
}

int gdr_k7_tune_get_rounds(int32_t N, int32_t H, int32_t W, int32_t V, int32_t kind, int32_t* chosen, int32_t* rounds_done,
                           float* us4) {
    std::lock_guard<std::mutex> lk(g_k7_mu);
    K7Tune* best = k7_latest(N, H, W, V, kind);
    if (!best) { set_error("k7_tune_get_rounds: no launch of this shape yet", hipSuccess); return GDR_ERR_INVALID_ARG; }
    k7_harvest(*best);
    if (chosen) *chosen = best->decided ? best->chosen : -1;
    if (rounds_done) *rounds_done = best->rounds_done;
    if (us4) for (int r = 0; r < 2; ++r) for (int v = 0; v < 2; ++v) us4[2 * r + v] = best->us_round[r][v];
    return GDR_OK;
}

int gdr_k7_tune_force_first(int32_t N, int32_t H, int32_t W, int32_t V, int32_t kind, int32_t variant) {
    std::lock_guard<std::mutex> lk(g_k7_mu);
    K7Tune* best = k7_latest(N, H, W, V, kind);
    if (!best) { set_error("k7_tune_force_first: no launch of this shape yet", hipSuccess); return GDR_ERR_INVALID_ARG; }
    k7_harvest(*best);
    if (best->rounds_done >= 2) { set_error("k7_tune_force_first: the shape's choice is already final", hipSuccess); return GDR_ERR_INVALID_ARG; }
    best->chosen = variant ? 1 : 0;
    best->decided = true;
    if (best->rounds_done == 0) { best->rounds_done = 1; best->us[0] = best->us[1] = 0.f; best->got = 0; }
    return GDR_OK;
}

void gdr_view_history_reset(void) {
    {   // (the K7 choices start over as well; events of measurements in flight are left to finish)
        std::lock_guard<std::mutex> lk7(g_k7_mu);
        for (auto& kv : g_k7) {
            kv.second.calls = 0; kv.second.chosen = 0; kv.second.got = 0; kv.second.us[0] = kv.second.us[1] = 0.f;
            kv.second.decided = false; kv.second.last_use = 0; kv.second.rounds_done = 0;
            for (int r = 0; r < 2; ++r) kv.second.us_round[r][0] = kv.second.us_round[r][1] = 0.f;
        }
    }
    std::lock_guard<std::mutex> lk(g_hist_mu);
    for (auto& kv : g_hist) {     // (the pinned report words stay: a kernel in flight may still write them)
        kv.second.d_per_n = 0.0; kv.second.n_long = kv.second.n_medium = 0; kv.second.deep_ttl = 0; kv.second.reported = false;
        if (kv.second.stats) for (int k = 0; k < kStatViews * 4; ++k) kv.second.stats[k] = 0xFFFFFFFFu;
    }
}

}  // extern "C"

// =================================================================================
// 2DGS surfel path (include/gsr.h)
// =================================================================================
namespace gdr {
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

// one forward call per view for the surfel boundary (see gdr_forward_view)
int gsr_forward_view(const gdr_settings* s, const gsr_inputs* in, const gdr_view_plan* plan, void* workspace,
                     const gdr_view_opts* opts, const gdr_same_as* same, const gsr_outputs* out, gdr_view_state* state,
                     void* stream) {
    int rc = check_surfel(s, in);
    if (rc) return rc;
    if (!plan || !out || !state || !out->color || !out->allmap || (in->N > 0 && !out->radii)) {
        set_error("surfel forward_view: NULL argument", hipSuccess);
        return GDR_ERR_INVALID_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    return forward_view_impl(
        s, in->N, 1, plan, workspace, opts, same, out->radii, state, st,
        [&](const gdr_geom& g) -> int {
            hipError_t e = launch_surfel_preprocess_fwd(s, in, &g, out->radii, st);
            if (e != hipSuccess) return hip_fail("surfel_preprocess_fwd", e);
            return debug_sync(s, "surfel_preprocess_fwd", st);
        },
        [&](const gdr_geom& g, const gdr_binning& b, const gdr_image& im) -> int {
            return gsr_composite_forward(s, &g, &b, &im, out, stream);
        });
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
    { K7Scope k7((int)in->N, s->image_height, s->image_width, 1, 3, bin->k7_class, st);
      e = launch_surfel_render_bwd(s, geom, bin, img, gin, gout->scratch, st); }
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
    { K7Scope k7(N, s->image_height, s->image_width, 1, 3, bin->k7_class, st);
      e = launch_surfel_render_bwd(s, geom, bin, img, gin, grad_rec, st); }
    if (e != hipSuccess) return hip_fail("surfel_render_bwd", e);
    return debug_sync(s, "surfel_render_bwd", st);
}

int gsr_render_backward_views(int32_t V, const gdr_settings* s, int32_t N, const gdr_geom* geoms, const gdr_binning* bins,
                              const gdr_image* imgs, const gsr_grad_inputs* gins, float* const* grad_recs,
                              int32_t interleave, void* stream) {
    int rc = check_bwd_views(V, s, geoms, bins, imgs, "surfel render_backward_views: NULL argument");
    if (rc) return rc;
    if (N <= 0) return GDR_OK;
    if (!gins || !grad_recs) { set_error("surfel render_backward_views: NULL argument", hipSuccess); return GDR_ERR_INVALID_ARG; }
    hipStream_t st = (hipStream_t)stream;
    for (int v = 0; v < V; ++v) {
        if (!gins[v].dL_dcolor || !grad_recs[v]) { set_error("surfel render_backward_views: NULL view buffer", hipSuccess); return GDR_ERR_INVALID_ARG; }
        if (!bins[v].grad_rec_cleared) {
            hipError_t e = hipMemsetAsync(grad_recs[v], 0, (size_t)N * GSR_GRAD_FLOATS * sizeof(float), st);
            if (e != hipSuccess) return hip_fail("memset gradient records", e);
        }
    }
    hipError_t e;
    { K7Scope k7(N, s[0].image_height, s[0].image_width, V, 3, k7_class_views(V, bins), st);
      e = launch_surfel_render_bwd_views(V, s, geoms, bins, imgs, gins, grad_recs, interleave, st); }
    if (e != hipSuccess) return hip_fail("surfel_render_bwd_views", e);
    return debug_sync(&s[0], "surfel_render_bwd_views", st);
}

int gsr_means2d_of_view(const gdr_settings* s, int32_t N, const gdr_geom* geom, const int32_t* radii,
                        const float* grad_rec, float* dL_dmean2D, void* stream) {
    if (N <= 0) return GDR_OK;
    if (!s || !geom || !radii || !grad_rec || !dL_dmean2D) { set_error("means2d_of_view: NULL argument", hipSuccess); return GDR_ERR_INVALID_ARG; }
    hipError_t e = launch_surfel_means2d_view(N, s, geom, radii, grad_rec, dL_dmean2D, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail("surfel_means2d_view", e);
    return GDR_OK;
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
