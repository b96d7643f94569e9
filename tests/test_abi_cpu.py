"""CPU: the C-ABI shared library loads, exports every symbol include/gdr.h declares, the
size/carve helpers work without a GPU, and the product path fails loudly off-GPU
(no oracle / CPU fallback anywhere in the product packages)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for hdr in ("gdr.h", "gsr.h"):   # every header under include/
        src = open(os.path.join(ROOT, "include", hdr)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(g[ds]r_[a-z_0-9]+)\s*\(", src))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from generativedensification_amd import _lib as L

    lib = L.load()
    names = _declared_symbols()
    assert len(names) >= 23 and "gsr_backward" in names and sorted(os.listdir(os.path.join(ROOT, "include"))) == ["gdr.h", "gsr.h"]
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/*.h but not exported"
        assert n in L.EXPORTED_SYMBOLS, f"{n} has no ctypes prototype"
    assert lib.gdr_abi_version() == L.ABI_VERSION == 17
    assert lib.gdr_build_tag() == b"release"


def test_workspace_sizes_and_carving_without_gpu():
    from generativedensification_amd import _lib as L

    lib = L.load()
    prev = 0
    for n in (0, 1, 255, 256, 257, 100_000, 2_000_000):
        b = lib.gdr_geom_bytes(n)
        assert b >= prev and b % 256 == 0
        prev = b
    assert lib.gdr_geom_bytes(2_000_000) >= 2_000_000 * (4 + 64 + 24 + 16 + 4 + 1)
    assert lib.gdr_binning_bytes(3_000_000) >= 3_000_000 * 24
    assert lib.gdr_image_bytes(800, 800) >= 2500 * 8 + 640000 * 8
    # carving a fake (never dereferenced) base address: pointers are ordered, aligned, in range
    g = L.GdrGeom()
    base = 0x10000000
    assert lib.gdr_geom_carve(C.c_void_p(base), 1000, C.byref(g)) == 0
    ptrs = [g.depths, g.rec, g.cov3D, g.rect, g.tiles_touched, g.clamped, g.block_sums, g.block_offs, g.num_rendered]
    assert ptrs == sorted(ptrs) and all(p % 256 == 0 for p in ptrs)
    assert ptrs[0] == base and ptrs[-1] + 4 <= base + lib.gdr_geom_bytes(1000)
    assert lib.gdr_geom_carve(C.c_void_p(base + 4), 1000, C.byref(g)) == -1  # unaligned -> GDR_ERR_INVALID_ARG
    assert b"unaligned" in lib.gdr_last_error()
    b = L.GdrBinning()
    assert lib.gdr_binning_carve(C.c_void_p(base), 5000, C.byref(b)) == 0
    assert b.keys[1] - b.keys[0] >= 5000 * 8 and b.values[1] - b.values[0] >= 5000 * 4


def test_surfel_workspace_sizes_and_errors_without_gpu():
    from generativedensification_amd import _lib as L

    lib = L.load()
    assert lib.gsr_geom_bytes(1_000_000) >= 1_000_000 * (4 + 96 + 16 + 4 + 1) and lib.gsr_geom_bytes(1000) % 256 == 0
    assert lib.gsr_image_bytes(800, 800) >= 2500 * 12 + 640000 * (8 + 12)
    g, im = L.GdrGeom(), L.GdrImage()
    base = 0x20000000
    assert lib.gsr_geom_carve(C.c_void_p(base), 1000, C.byref(g)) == 0 and g.cov3D is None
    assert g.rect - g.rec >= 1000 * 96
    assert lib.gsr_image_carve(C.c_void_p(base), 64, 48, C.byref(im)) == 0
    assert im.final_T - im.n_contrib >= 2 * 64 * 48 * 4 and im.tile_order - im.final_T >= 3 * 64 * 48 * 4
    assert lib.gsr_geom_carve(C.c_void_p(base + 8), 10, C.byref(g)) == -1
    s, i = L.GdrSettings(), L.GsrInputs()
    assert lib.gsr_preprocess_forward(None, None, None, None, None, None) == -1
    s.image_height, s.image_width = 64, 64
    assert lib.gsr_preprocess_forward(C.byref(s), C.byref(i), C.byref(g), None, None, None) == -1
    assert lib.gsr_backward(C.byref(s), C.byref(i), None, None, None, 0, None, None, None, None) == -1


def test_surfel_product_path_has_no_cpu_fallback():
    import diff_surfel_rasterization as D

    rs = D.GaussianRasterizationSettings(image_height=32, image_width=32, tanfovx=0.4, tanfovy=0.4, bg=torch.ones(3),
                                         scale_modifier=1.0, viewmatrix=torch.eye(4), projmatrix=torch.eye(4),
                                         sh_degree=0, campos=torch.zeros(3), prefiltered=False, debug=False)
    r = D.GaussianRasterizer(rs)
    n = 10
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r(means3D=torch.zeros(n, 3), means2D=torch.zeros(n, 4), opacities=torch.ones(n, 1), shs=torch.zeros(n, 1, 3),
          scales=torch.ones(n, 2), rotations=torch.ones(n, 4))
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=torch.zeros(n, 3), means2D=torch.zeros(n, 4), opacities=torch.ones(n, 1), scales=torch.ones(n, 2),
          rotations=torch.ones(n, 4))


def test_argument_errors_are_reported_not_thrown():
    from generativedensification_amd import _lib as L

    lib = L.load()
    s, i, g = L.GdrSettings(), L.GdrInputs(), L.GdrGeom()
    assert lib.gdr_preprocess_forward(None, None, None, None, None, None) == -1
    s.image_height, s.image_width = 64, 64
    assert lib.gdr_preprocess_forward(C.byref(s), C.byref(i), C.byref(g), None, None, None) == -1  # bg/view/proj NULL
    assert lib.gdr_mark_visible(-1, None, None, None, None, None) == -1
    # size limit of the build: N > GDR_MAX_GAUSSIANS is refused before any device work (GDR_ERR_UNSUPPORTED)
    i.N = (1 << 27) + 1
    s.bg = s.viewmatrix = s.projmatrix = 0x1000
    assert lib.gdr_preprocess_forward(C.byref(s), C.byref(i), C.byref(g), None, None, None) == -3
    assert b"GDR_MAX_GAUSSIANS" in lib.gdr_last_error()
    j = L.GsrInputs()
    j.N = (1 << 27) + 1
    assert lib.gsr_preprocess_forward(C.byref(s), C.byref(j), C.byref(g), None, None, None) == -3


def test_product_path_has_no_cpu_fallback():
    import diff_gaussian_rasterization as D

    rs = D.GaussianRasterizationSettings(image_height=32, image_width=32, tanfovx=0.4, tanfovy=0.4, bg=torch.ones(3),
                                         scale_modifier=1.0, viewmatrix=torch.eye(4), projmatrix=torch.eye(4),
                                         sh_degree=0, campos=torch.zeros(3), prefiltered=False, debug=False)
    r = D.GaussianRasterizer(rs)
    n = 10
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r(means3D=torch.zeros(n, 3), means2D=torch.zeros(n, 4), opacities=torch.ones(n, 1), shs=torch.zeros(n, 1, 3),
          scales=torch.ones(n, 3), rotations=torch.ones(n, 4))
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=torch.zeros(n, 3), means2D=torch.zeros(n, 4), opacities=torch.ones(n, 1), scales=torch.ones(n, 3),
          rotations=torch.ones(n, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=torch.zeros(n, 3), means2D=torch.zeros(n, 4), opacities=torch.ones(n, 1), shs=torch.zeros(n, 1, 3))


def test_product_packages_never_touch_the_oracle():
    for pkg in ("generativedensification_amd", "diff_gaussian_rasterization", "diff_surfel_rasterization", "simple_knn"):
        for dp, _, files in os.walk(os.path.join(ROOT, pkg)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                    txt = open(os.path.join(dp, f)).read()
                    assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), (dp, f)
                    assert "gdr_oracle" not in txt.replace("oracle/gdr_oracle.c", "").replace("oracle_bin()", ""), (dp, f)
                    assert "gsr_oracle" not in txt.replace("oracle/gsr_oracle.c", ""), (dp, f)


def test_settings_record_has_the_reference_fields_in_order():
    from diff_gaussian_rasterization import GaussianRasterizationSettings as S

    assert S._fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                         "projmatrix", "sh_degree", "campos", "prefiltered", "debug")  # lightning/renderer.py:111-124


def test_host_policy_helpers_without_gpu():
    """Host-side policy of the multi-view nodes: side-stream count by tile count, its test override, and the
    per-call segment-length choice of the cut lists."""
    from generativedensification_amd import _lib as L
    from generativedensification_amd import rasterizer as R

    saved = R.K.BIN_STREAM, R.K.SEG_LEN
    try:
        R.K.BIN_STREAM = None
        assert R.side_count(800, 800) == 2 and R.side_count(512, 512) == 3 and R.side_count(1080, 1920) == 2
        R.K.BIN_STREAM = 0
        assert R.side_count(800, 800) == 0
        R.K.BIN_STREAM = 4
        assert R.side_count(64, 64) == 4
        R.K.SEG_LEN = None
        assert R._seg_len_for(10_000) == 256
        # the automatic choice: 512 on large images whose tiles are all busy with long lists, 256 otherwise
        for D, tiles, busy, want in ((3_400_000, 2500, None, 512), (3_400_000, 2500, 2000, 512), (3_400_000, 2500, 330, 256),
                                     (730_000, 2500, 2500, 256), (1_070_000, 1024, 1024, 256)):
            assert R._seg_len_for(D, tiles, busy) == want, (D, tiles, busy)
        R.K.SEG_LEN = 4096
        assert R._seg_len_for(10_000) == 4096
        R.K.SEG_LEN = 700           # rounded down to a multiple of 256
        assert R._seg_len_for(10_000) == 512
        R.K.SEG_LEN = 0             # lists are never cut
        assert R._seg_len_for(3_400_000, 2500, None) == 0
    finally:
        R.K.BIN_STREAM, R.K.SEG_LEN = saved
    lib = L.load()
    # the cut-list tables are carved for the segment length the caller is going to use (80 / 40 / 0 bytes per duplicate)
    b256, b512, b0 = (lib.gdr_binning_bytes_for(4_000_000, sl, 0, 0) for sl in (256, 512, 0))
    assert b256 == lib.gdr_binning_bytes(4_000_000) and b256 - b512 >= 4_000_000 * 39 and b512 - b0 >= 4_000_000 * 39
    bb = L.GdrBinning()
    assert lib.gdr_binning_carve_for(C.c_void_p(0x10000000), 4_000_000, 512, 0, 0, C.byref(bb)) == 0
    assert bb.seg_len == 512 and bb.seg_cap == 4_000_000 // 512 + 1 and not bb.tile_hist and bb.hist_width == 0
    assert lib.gdr_binning_carve_for(C.c_void_p(0x10000000), 4_000_000, 0, 0, 0, C.byref(bb)) == 0 and bb.seg_len == 0 and bb.seg_cap == 0
    # the count matrix of the direct tile binning: min(256, N / 1024) rows of tiles words (+ a totals row); none beyond 16384 tiles
    assert lib.gdr_binning_carve_for(C.c_void_p(0x10000000), 4_000_000, 256, 2_000_000, 2500, C.byref(bb)) == 0
    assert bb.tile_hist and bb.hist_width == 256
    assert lib.gdr_binning_bytes_for(4_000_000, 256, 2_000_000, 2500) - b256 >= 2500 * 257 * 4
    assert lib.gdr_binning_carve_for(C.c_void_p(0x10000000), 4_000_000, 256, 100_000, 2500, C.byref(bb)) == 0 and bb.hist_width == 98
    assert lib.gdr_binning_carve_for(C.c_void_p(0x10000000), 4_000_000, 256, 2_000_000, 32_400, C.byref(bb)) == 0 and not bb.tile_hist
    small, big = lib.gdr_binning_bytes(1000), lib.gdr_binning_bytes(4_000_000)
    assert big > small and big >= 4_000_000 * (8 + 8 + 4 + 4 + 8) + 2 * (4_000_000 // 2048) * 10 * 256 * 4



def test_view_plan_and_shape_history_without_gpu():
    """gdr_view_plan_for / gdr_view_history_* (include/gdr.h, v14): sizing of the one allocation of a native forward call
    from the per-shape history the library keeps; no device work."""
    import ctypes as C
    from generativedensification_amd import _lib as L
    lib = L.load()
    N, H, W = 50_000, 160, 208
    lib.gdr_view_history_reset()
    plan = L.GdrViewPlan()
    assert lib.gdr_view_plan_for(N, H, W, 0, 0, None, C.byref(plan)) == 0
    assert plan.have_binning == 0 and plan.deferred == 0 and plan.capacity == 0      # no history: K1 + read-back first
    fixed = plan.bytes
    assert fixed >= lib.gdr_geom_bytes(N) + lib.gdr_image_bytes(H, W)
    assert lib.gdr_view_plan_for(N, H, W, 0, 123_456, None, C.byref(plan)) == 0      # exactly sized
    assert plan.have_binning == 1 and plan.deferred == 0 and plan.capacity == 123_456 and plan.seg_len == 256
    assert plan.bytes > fixed
    lib.gdr_view_history_set(N, H, W, 0, 2.0)
    assert lib.gdr_view_history_get(N, H, W, 0) == 2.0 and lib.gdr_view_history_get(N, H, W, 1) == 0.0   # (surfel: own history)
    assert lib.gdr_view_history_get(40_000, H, W, 0) == 2.0        # same power-of-two bucket of N
    assert lib.gdr_view_plan_for(N, H, W, 0, 0, None, C.byref(plan)) == 0
    assert plan.deferred == 1 and plan.have_binning == 1 and plan.capacity == int(2.0 * N * 1.5) + 4096
    opts = L.GdrViewOpts(1024, -1, -1, 0, 1, 0)                     # segment length override, radix partition (no count matrix)
    big = plan.bytes
    assert lib.gdr_view_plan_for(N, H, W, 0, 0, C.byref(opts), C.byref(plan)) == 0
    assert plan.seg_len == 1024 and plan.bytes < big
    # 800x800 with long lists everywhere: the policy raises the segment length to 512
    lib.gdr_view_history_set(2_000_000, 800, 800, 0, 1.8)
    assert lib.gdr_view_plan_for(2_000_000, 800, 800, 0, 0, None, C.byref(plan)) == 0 and plan.seg_len == 512
    lib.gdr_view_history_reset()
    assert lib.gdr_view_history_get(N, H, W, 0) == 0.0
    assert lib.gdr_view_plan_for(-1, H, W, 0, 0, None, C.byref(plan)) == -1
    # the K7 choice of a shape (v15): nothing launched yet -> an error with a message, never a default; the override is a
    # process-wide setting that takes any value (-1 / 0 / 1) without device work
    ch, u0, u1 = C.c_int32(-7), C.c_float(-1), C.c_float(-1)
    assert lib.gdr_k7_tune_get(N, H, W, 1, 0, C.byref(ch), C.byref(u0), C.byref(u1)) == -1
    assert b"k7_tune_get" in lib.gdr_last_error() and ch.value == -7
    for mode in (0, 1, -1):
        lib.gdr_k7_tune_override(mode)


def test_view_reuse_probe_argument_checks_and_host_half_without_a_gpu():
    """include/gdr.h gdr_view_reuse_probe (v16): bad arguments are errors with a message; with nothing to compare on the
    device (no candidate whose host scalars match, no same-as pair) the call answers -1 / 0 without touching the GPU."""
    import ctypes as C
    from generativedensification_amd import _lib as L
    lib = L.load()
    fake = 0x1000                       # never dereferenced on this path
    s = L.GdrSettings(64, 64, 0.4, 0.4, 1.0, 1, 0, 0, fake, fake, fake, fake)
    other = L.GdrSettings(64, 80, 0.4, 0.4, 1.0, 1, 0, 0, fake, fake, fake, fake)       # another image width: host half differs
    cands = (L.GdrSettings * 2)(other, L.GdrSettings(64, 64, 0.4, 0.4, 1.0, 3, 0, 0, fake, fake, fake, fake))
    match, differ = C.c_int32(7), C.c_uint32(9)
    assert lib.gdr_view_reuse_probe(None, 0, None, None, fake, C.byref(match), C.byref(differ), None) == -1
    assert b"view_reuse_probe" in lib.gdr_last_error()
    assert lib.gdr_view_reuse_probe(C.byref(s), L.GDR_REUSE_MAX + 1, cands, None, fake, C.byref(match), C.byref(differ), None) == -1
    assert lib.gdr_view_reuse_probe(C.byref(s), 2, cands, None, None, C.byref(match), C.byref(differ), None) == -1     # no scratch
    assert lib.gdr_view_reuse_probe(C.byref(s), 2, cands, None, fake, C.byref(match), C.byref(differ), None) == 0
    assert match.value == -1 and differ.value == 0
    assert lib.gdr_view_reuse_probe(C.byref(s), 0, None, None, fake, C.byref(match), C.byref(differ), None) == 0 and match.value == -1


def test_debug_knobs_map_to_view_opts():
    from generativedensification_amd import rasterizer as R
    K = R.K
    saved = K.SEG_LEN, K.DEEP_MAX_BUSY, K.DEEP_MIN_MEAN, K.FORCE_GLOBAL_SORT, K.FORCE_RADIX_PARTITION, K.LAUNCH_HINTS
    try:
        K.SEG_LEN = K.DEEP_MAX_BUSY = K.DEEP_MIN_MEAN = None
        K.FORCE_GLOBAL_SORT = K.FORCE_RADIX_PARTITION = False
        K.LAUNCH_HINTS = True
        o = R._view_opts()
        assert (o.seg_len, o.deep_max_busy, o.deep_min_mean, o.global_sort, o.radix_partition, o.no_hints) == (-1, -1, -1, 0, 0, 0)
        K.SEG_LEN, K.DEEP_MAX_BUSY, K.DEEP_MIN_MEAN, K.FORCE_GLOBAL_SORT, K.LAUNCH_HINTS = 700, 0, 4000, True, False
        o = R._view_opts()
        assert (o.seg_len, o.deep_max_busy, o.deep_min_mean, o.global_sort, o.radix_partition, o.no_hints) == (512, 0, 4000, 1, 0, 1)
    finally:
        K.SEG_LEN, K.DEEP_MAX_BUSY, K.DEEP_MIN_MEAN, K.FORCE_GLOBAL_SORT, K.FORCE_RADIX_PARTITION, K.LAUNCH_HINTS = saved


def test_gaussian_rasterizer_is_a_real_module_set_up_on_first_use():
    """The reference builds a new GaussianRasterizer per render call (lightning/renderer.py:106-126); ours defers
    nn.Module's set-up until something touches the module machinery — and must then behave like any nn.Module."""
    import copy
    import pickle
    import torch
    from torch import nn
    import diff_gaussian_rasterization as D
    import diff_surfel_rasterization as DS
    rs = D.GaussianRasterizationSettings(8, 8, 1.0, 1.0, torch.ones(3), 1.0, torch.eye(4), torch.eye(4), 1, torch.zeros(3), False, False)
    for M in (D.GaussianRasterizer, DS.GaussianRasterizer):
        m = M(raster_settings=rs)
        assert isinstance(m, nn.Module) and m.raster_settings is rs and "_parameters" not in m.__dict__
        assert "GaussianRasterizer" in repr(m) and "_parameters" in m.__dict__ and m.raster_settings is rs
        m2 = M(rs).to("cpu").eval()
        assert m2.training is False and list(m2.parameters()) == [] and m2.state_dict() == {} and m2.raster_settings is rs
        m3 = M(rs)
        m3.register_forward_pre_hook(lambda mod, a: None)
        m3.extra = 3
        assert m3.extra == 3 and m3.raster_settings is rs
        assert len(list(nn.Sequential(M(rs)).modules())) == 2
        assert isinstance(copy.deepcopy(M(rs)), M) and isinstance(pickle.loads(pickle.dumps(M(rs))), M)
        with pytest.raises(Exception, match="excatly one"):         # the untouched instance's call goes straight to forward
            M(rs)(means3D=torch.zeros(1, 3), means2D=torch.zeros(1, 4), opacities=torch.zeros(1, 1))
        with pytest.raises(AttributeError):
            M(rs).no_such_attribute
