"""Shared helpers of the parity tests: build a seeded case, run it through the oracle
(checker) and through the HIP product path, compare."""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from generativedensification_amd.camera import orbit_cameras
from generativedensification_amd.synthetic import make_scene
from oracle.gdr_oracle import Oracle, Settings

FOV = 0.75


def make_case(N, H, W, seed, deg=3, sigma0=(0.0052, 0.00065), cam_index=1, n_cams=4, bg=(1.0, 1.0, 1.0),
              scale_modifier=1.0, colors_precomp=False, cov_precomp=False):
    sc = make_scene(N, seed, sh_degree=deg, sigma0=sigma0)
    cam = orbit_cameras(n_cams, W, H)[cam_index]
    case = dict(
        N=N, H=H, W=W, deg=deg,
        means3D=sc["centers"].contiguous(),
        opacities=torch.sigmoid(sc["opacity"]).contiguous(),
        shs=None if colors_precomp else sc["shs"].contiguous(),
        colors_precomp=torch.sigmoid(sc["shs"][:, 0, :]).contiguous() if colors_precomp else None,
        scales=torch.exp(sc["scales"]).contiguous(),
        rotations=torch.nn.functional.normalize(sc["rotations"]).contiguous(),
        cov3D_precomp=None,
        view=cam.world_view_transform.contiguous(), proj=cam.full_proj_transform.contiguous(),
        campos=cam.camera_center.contiguous(), bg=torch.tensor(bg, dtype=torch.float32),
        tanfovx=math.tan(FOV * 0.5), tanfovy=math.tan(FOV * 0.5), scale_modifier=scale_modifier,
    )
    if cov_precomp:
        o = Oracle("f32")
        tmp = o.forward(case["means3D"].numpy(), case["opacities"].numpy(), settings_np(case),
                        shs=None if colors_precomp else case["shs"].numpy(),
                        colors_precomp=case["colors_precomp"].numpy() if colors_precomp else None,
                        scales=case["scales"].numpy(), rotations=case["rotations"].numpy())
        # covariance of every Gaussian (also the culled ones): recompute densely in torch
        from oracle.torch_ref import quat_to_R
        R = quat_to_R(case["rotations"])
        Mm = R * (scale_modifier * case["scales"])[:, None, :]
        S = Mm @ Mm.transpose(1, 2)
        case["cov3D_precomp"] = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).contiguous()
        case["scales"] = None
        case["rotations"] = None
        del tmp
    return case


def settings_np(case) -> Settings:
    return Settings(case["H"], case["W"], case["tanfovx"], case["tanfovy"], case["bg"].numpy(),
                    case["scale_modifier"], case["view"].numpy(), case["proj"].numpy(), case["deg"],
                    case["campos"].numpy(), False, False)


def _np(t):
    return None if t is None else t.detach().cpu().numpy()


def run_oracle(case, precision="f32", grads=None, nthreads=1):
    o = Oracle(precision, nthreads=nthreads)
    out = o.forward(_np(case["means3D"]), _np(case["opacities"]), settings_np(case), shs=_np(case["shs"]),
                    colors_precomp=_np(case["colors_precomp"]), scales=_np(case["scales"]),
                    rotations=_np(case["rotations"]), cov3D_precomp=_np(case["cov3D_precomp"]))
    g = None
    if grads is not None:
        g = o.backward(out, _np(grads[0]), _np(grads[1]), _np(grads[2]))
    return out, g


def settings_torch(case, dev):
    from generativedensification_amd.rasterizer import GaussianRasterizationSettings

    return GaussianRasterizationSettings(
        image_height=case["H"], image_width=case["W"], tanfovx=case["tanfovx"], tanfovy=case["tanfovy"],
        bg=case["bg"].to(dev), scale_modifier=case["scale_modifier"], viewmatrix=case["view"].to(dev),
        projmatrix=case["proj"].to(dev), sh_degree=case["deg"], campos=case["campos"].to(dev),
        prefiltered=False, debug=False)


def run_hip(case, grads=None, dev="cuda:0"):
    """Through the product C ABI (generativedensification_amd.rasterizer -> libgdr_hip.so)."""
    from generativedensification_amd import rasterizer as R

    dev = torch.device(dev)
    rs = settings_torch(case, dev)
    e = torch.empty(0, device=dev)
    t = lambda k: e if case[k] is None else case[k].to(dev)
    color, radii, depth, alpha, st, keep = R.forward_raw(
        t("means3D"), t("shs"), t("colors_precomp"), t("opacities"), t("scales"), t("rotations"),
        t("cov3D_precomp"), rs)
    torch.cuda.synchronize()
    out = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in st.tensors().items()}
    out.update(color=color.cpu().numpy(), radii=radii.cpu().numpy(), depth=depth.cpu().numpy(),
               alpha=alpha.cpu().numpy())
    g = None
    if grads is not None:
        gg = R.backward_raw(st, keep, rs, radii, grads[0].to(dev), grads[1].to(dev), grads[2].to(dev))
        torch.cuda.synchronize()
        g = {k: (None if v is None else v.cpu().numpy()) for k, v in gg.items()}
    return out, g


def rand_grads(case, seed=123):
    g = torch.Generator().manual_seed(seed)
    H, W = case["H"], case["W"]
    return (torch.randn(3, H, W, generator=g), torch.randn(1, H, W, generator=g),
            torch.randn(1, H, W, generator=g))


def rel_inf(a, b):
    """||a-b||_inf / max(||b||_inf, tiny)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)) if a.size else 0.0


def outlier_fraction(a, b, rtol=1e-4, atol=1e-4):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float((np.abs(a - b) > atol + rtol * np.abs(b)).mean()) if a.size else 0.0


# The parity line of the images, held where it IS (round-5 verdict, next #5).  HIP and the f32 oracle evaluate alpha with two
# valid fp32 programs (v_exp_f32 on a log2e-prescaled conic vs glibc expf), so on a handful of pixels per image one
# contributor of weight ~1/255 is kept by one and skipped by the other (alpha on the other side of 1/255, or T on the other
# side of 1e-4): measured per 800x800 image n_contrib differs on 0-1 pixels, final T on 0-2, colour on 0-1.  The bars are
# COUNTS a little above that — not fractions of the image (1e-4 of 640 000 pixels = 64: a regression that flips 50 pixels
# per image passed until round 5).
MAX_NCONTRIB_PIXELS, MAX_FINAL_T_PIXELS, MAX_IMAGE_PIXELS = 4, 8, 4


def image_parity_counts(h, o, mask=None):
    """(pixels whose contributor count differs, pixels whose final T is outside, pixels with colour / depth / alpha outside
    the per-element bar, PSNR of the clamped colour).  h / o: dicts with color (3,H,W), depth, alpha (1,H,W), n_contrib,
    final_T (H,W); mask: optional boolean (H,W) restricting the comparison."""
    m = np.ones(h["n_contrib"].shape, bool) if mask is None else mask
    nc = int((h["n_contrib"].view(np.uint32)[m] != np.asarray(o["n_contrib"]).view(np.uint32)[m]).sum())
    a, b = np.asarray(h["final_T"], np.float64)[m], np.asarray(o["final_T"], np.float64)[m]
    ft = int((np.abs(a - b) > 1e-6 + 1e-4 * np.abs(b)).sum())
    bad = np.zeros(int(m.sum()), bool)
    for k in ("color", "depth", "alpha"):
        x, y = np.asarray(h[k], np.float64)[:, m], np.asarray(o[k], np.float64)[:, m]
        bad |= (np.abs(x - y) > 1e-5 + 1e-4 * np.abs(y)).any(axis=0)
    return nc, ft, int(bad.sum()), psnr(np.clip(h["color"][:, m], 0, 1), np.clip(o["color"][:, m], 0, 1))


def assert_image_parity(h, o, tag, mask=None):
    nc, ft, px, p = image_parity_counts(h, o, mask)
    for k in ("color", "depth", "alpha"):
        x, y = (h[k], o[k]) if mask is None else (h[k][:, mask], o[k][:, mask])
        assert rel_inf(x, y) < 5e-3, (tag, k)        # (one contributor of weight 1/255 on a pixel, no more)
    assert nc <= MAX_NCONTRIB_PIXELS and ft <= MAX_FINAL_T_PIXELS and px <= MAX_IMAGE_PIXELS and p > 60.0, \
        (tag, dict(n_contrib_pixels=nc, final_T_pixels=ft, image_pixels=px, psnr=p))
    return nc, ft, px, p


def psnr(a, b):
    mse = float(((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2).mean())
    return 10 * math.log10(1.0 / max(mse, 1e-30))


# ---- 2DGS surfel path -------------------------------------------------------------------------------------
def make_surfel_case(N, H, W, seed, **kw):
    """make_case with (N,2) scales (renderer_2dgs.py:92-96)."""
    case = make_case(N, H, W, seed, **kw)
    case["scales"] = case["scales"][:, :2].contiguous()
    case["transMat_precomp"] = None
    return case


def run_surfel_oracle(case, precision="f32", grads=None, nthreads=1):
    from oracle.gsr_oracle import SurfelOracle

    o = SurfelOracle(precision, nthreads=nthreads)
    out = o.forward(_np(case["means3D"]), _np(case["opacities"]), settings_np(case), shs=_np(case["shs"]),
                    colors_precomp=_np(case["colors_precomp"]), scales=_np(case["scales"]),
                    rotations=_np(case["rotations"]), transMat_precomp=_np(case.get("transMat_precomp")))
    g = o.backward(out, _np(grads[0]), _np(grads[1])) if grads is not None else None
    return out, g


def run_surfel_hip(case, grads=None, dev="cuda:0"):
    """Through the product C ABI (generativedensification_amd.surfel_rasterizer -> libgdr_hip.so, include/gsr.h)."""
    from generativedensification_amd import surfel_rasterizer as S

    dev = torch.device(dev)
    rs = settings_torch(case, dev)
    e = torch.empty(0, device=dev)
    t = lambda k: e if case.get(k) is None else case[k].to(dev)
    color, radii, allmap, st, keep = S.forward_raw(t("means3D"), t("shs"), t("colors_precomp"), t("opacities"),
                                                   t("scales"), t("rotations"), t("transMat_precomp"), rs)
    torch.cuda.synchronize()
    out = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in st.tensors().items()}
    out.update(color=color.cpu().numpy(), radii=radii.cpu().numpy(), allmap=allmap.cpu().numpy())
    g = None
    if grads is not None:
        gg = S.backward_raw(st, keep, rs, radii, grads[0].to(dev), None if grads[1] is None else grads[1].to(dev))
        torch.cuda.synchronize()
        g = {k: (None if v is None else v.cpu().numpy()) for k, v in gg.items()}
    return out, g


def rand_surfel_grads(case, seed=123):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(3, case["H"], case["W"], generator=g), torch.randn(7, case["H"], case["W"], generator=g)


# ---- per-element gradient comparison (round-2: replaces max-norm-relative checks) ----------------------------------
def elem_stats(a, ref, rtol=1e-4, atol_rel=1e-6):
    """Per-ELEMENT comparison |a - ref| <= rtol |ref| + atol, atol = atol_rel * max|ref| (an absolute floor two orders
    below the old max-norm bar: elements that are sums of cancelling fp32 terms cannot be relative-exact to
    themselves).  Returns (fraction of elements outside, worst |a-ref| / (rtol |ref| + atol), max-norm relative error)."""
    a = np.asarray(a, np.float64).reshape(-1)
    ref = np.asarray(ref, np.float64).reshape(-1)
    if ref.size == 0:
        return 0.0, 0.0, 0.0
    scale = max(float(np.abs(ref).max()), 1e-300)
    tol = rtol * np.abs(ref) + atol_rel * scale
    err = np.abs(a - ref)
    return float((err > tol).mean()), float((err / tol).max()), float(err.max() / scale)


def tile_mask(tiles, H, W):
    """Boolean (H,W) mask of the pixels of the given tile ids (y * grid_x + x, 16x16 tiles)."""
    gx = (W + 15) // 16
    m = np.zeros((H, W), bool)
    for t in np.asarray(tiles).reshape(-1):
        ty, tx = divmod(int(t), gx)
        m[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16] = True
    return m


def pick_tiles(ranges, n_long=16, n_rand=32, seed=0):
    """Sample of tile ids for a sampled-tile oracle comparison: the n_long longest lists + n_rand random non-empty."""
    r = np.asarray(ranges).astype(np.int64).reshape(-1, 2)
    length = r[:, 1] - r[:, 0]
    order = np.argsort(-length, kind="stable")
    long_ = order[:n_long]
    rest = order[n_long:][length[order[n_long:]] > 0]
    g = np.random.default_rng(seed)
    rnd = g.choice(rest, size=min(n_rand, rest.size), replace=False) if rest.size else np.empty(0, np.int64)
    return np.unique(np.concatenate([long_[length[long_] > 0], rnd])).astype(np.int32)


# Fraction of gradient elements allowed outside 1e-4 |ref| + 1e-6 max|ref| of the float32 oracle.  Measured on MI355X
# at every BASELINE size (tests/test_gpu_oracle_fullsize.py): 0 to 9e-6, max-norm relative 2e-7..2e-6 —
# K7's atomics and K9's fused multiply-adds only reorder / contract fp32 sums.
MAX_OUTSIDE = 1e-4


def assert_grads(hg, g64, g32, keys, what, max_outside=MAX_OUTSIDE, rtol=1e-4, atol_rel=1e-6, maxnorm=1e-4):
    """hg: HIP; g32: float32 oracle (the bar); g64: float64 oracle (printed next to it, and a second assertion).
    Per ELEMENT: |hip - f32 oracle| <= rtol |ref| + atol_rel max|ref| with at most `max_outside` of the elements
    outside, max-norm relative error < maxnorm; and HIP no further from float64 than the f32 oracle is (x 1.25).
    No escape hatch: round 2's `or closer to float64` clause was never taken (0 elements outside at every BASELINE
    size, profiles/r02_fullsize_parity.log) and is gone."""
    for k in keys:
        r32 = np.asarray(g32[k]).reshape(hg[k].shape)
        r64 = np.asarray(g64[k]).reshape(hg[k].shape)
        out, worst, maxn = elem_stats(hg[k], r32, rtol, atol_rel)
        o_h64, _, m_h64 = elem_stats(hg[k], r64, rtol, atol_rel)
        o_3264, _, m_3264 = elem_stats(r32, r64, rtol, atol_rel)
        print(f"[{what}] {k:10s} vs f32 oracle: outside {out:.2e} worst/tol {worst:.1f} max-norm rel {maxn:.2e} | "
              f"vs f64: hip {o_h64:.2e} / {m_h64:.2e}, f32 oracle {o_3264:.2e} / {m_3264:.2e}")
        assert np.isfinite(hg[k]).all(), (what, k)
        assert out < max_outside, (what, k, out, o_h64, o_3264)
        assert maxn < maxnorm, (what, k, maxn)
        # as accurate as the fp32 algorithm allows: no further from float64 than the f32 oracle (+ a tenth of the bar: with
        # the oracle's block-wise sums both sit at the 1e-6 level where neither is "closer" in any meaningful sense)
        assert m_h64 <= 1.25 * m_3264 + 1e-5 and o_h64 <= 1.25 * o_3264 + 1e-4, (what, k, m_h64, m_3264, o_h64, o_3264)


# ---- 2DGS: the same per-element comparison; what the fp32 2DGS formulation forces is stated here -------------------------
# The published 2DGS ray-splat intersection evaluates k = x Tw - Tu, l = y Tw - Tv per pixel in fp32: for a small surfel
# far from the image origin ~800 * 2 cancels against ~1600, the cross product cancels again and (u, v) divides by what is
# left.  Two fp32 evaluations that differ in ONE rounding there differ from each other, per element, about as much as each
# differs from float64 (the f32 ORACLE ITSELF misses float64 in 0.1-3 % of the elements at the 3DGS floor, table below).
# Rounds 1-3: K6s / K7s used fma for k, l and v_rcp_f32 / v_exp_f32; the bar needed an absolute floor of 1e-5 (ten times the
# 3DGS one), 1.5e-3 of the elements outside on small scenes, and `worst_factor` 3-4 on the whole-image tests (HIP's single
# worst element up to 2.5x further from float64 than the oracle's — or 3x closer: luck of one ill-conditioned surfel).
# Round 4: the intersection runs in the oracle's own operation order with a correctly rounded quotient
# (render_surfel.hip GSR_ORACLE_ORDER), so (u, v), rho and depth of every (pixel, surfel) pair are the oracle's bit for bit.
# Measured on MI355X (profiles/r04_surfel_stats.txt), fraction of elements outside rtol 1e-4 |ref| + atol_rel max|ref|:
#                                              atol_rel 1e-6      3e-6       1e-5      | max-norm hip-f32 | hip-f64 / f32-f64
#   C5 sampled tiles (single call)              0                 0          0         | 1.2e-6           | 1.00
#   C5 whole image (single call)                3.5e-6            3.5e-6     2.0e-6    | 2.5e-5           | 1.00
#   small random scenes (single call)           5.0e-5            0          0         | 5.6e-6           | <= 1.75 (at 1e-6 level)
#   render_views, RAW inputs, 20 k surfels      3.0e-4            1.2e-4     3.8e-5    | 5.1e-5           | 1.00
#   render_views, RAW inputs, C5                3.9e-4            2.1e-4     7.8e-5    | 8.1e-4           | 1.00
#   (f32 oracle vs float64, same scenes:        8e-4 .. 3e-2      3e-4..1e-2)
# The RAW entries (multi-view node: sigmoid / exp / normalize inside K1s) start from activations that differ from torch's in
# the last bit, which the ill-conditioned geometry amplifies — they keep the 1e-5 floor; everything fed activated tensors
# meets the 3DGS bar at three times the 3DGS floor.  Asserted:
#   (a) vs the f32 oracle: fraction outside < max_outside — MAX_OUTSIDE = 1e-4 at atol_rel 3e-6 (single calls), 2e-4 at
#       atol_rel 1e-5 for the RAW multi-view entries;
#   (b) vs float64: HIP's fraction outside <= 1.25 x the f32 oracle's own + 5e-4;
#   (c) max-norm: HIP no further from the f32 oracle than 2 x the oracle's own distance from float64 (+1e-4), and no
#       further from float64 than worst_factor = 1.25 x the oracle's (+1e-5) or the north-star's absolute 1e-4.
SURFEL_ATOL_REL = 3e-6
SURFEL_MAX_OUTSIDE = MAX_OUTSIDE
SURFEL_RAW_ATOL_REL = 1e-5        # entries that take RAW tensors (activations inside K1s)
SURFEL_RAW_MAX_OUTSIDE = 2e-4


def assert_grads_surfel(hg, g64, g32, keys, what, max_outside=SURFEL_MAX_OUTSIDE, rtol=1e-4, atol_rel=SURFEL_ATOL_REL,
                        worst_factor=1.25):
    for k in keys:
        r32 = np.asarray(g32[k]).reshape(hg[k].shape)
        r64 = np.asarray(g64[k]).reshape(hg[k].shape)
        out, worst, maxn = elem_stats(hg[k], r32, rtol, atol_rel)
        o_h64, _, m_h64 = elem_stats(hg[k], r64, rtol, atol_rel)
        o_3264, _, m_3264 = elem_stats(r32, r64, rtol, atol_rel)
        print(f"[{what}] {k:10s} vs f32 oracle: outside {out:.2e} worst/tol {worst:.1f} max-norm rel {maxn:.2e} | "
              f"vs f64: hip {o_h64:.2e} / {m_h64:.2e}, f32 oracle {o_3264:.2e} / {m_3264:.2e}")
        if os.environ.get("GDR_TEST_STATS"):     # (scripts: the same three fractions at other absolute floors — printed on
            for ar in (1e-6, 3e-6, 3e-5, 1e-4):  #  top of the assertions below, which always run)
                print(f"[{what}] {k:10s}   atol_rel {ar:.0e}: hip-f32 {elem_stats(hg[k], r32, rtol, ar)[0]:.2e} hip-f64 "
                      f"{elem_stats(hg[k], r64, rtol, ar)[0]:.2e} f32-f64 {elem_stats(r32, r64, rtol, ar)[0]:.2e}")
        assert np.isfinite(hg[k]).all(), (what, k)
        few = 2.01 / max(r32.size, 1)                                        # two elements of a small array
        assert out < max(max_outside, few), (what, k, "(a)", out)
        assert o_h64 <= 1.25 * o_3264 + max(5e-4, few), (what, k, "(b)", o_h64, o_3264)
        assert maxn <= max(2.0, worst_factor) * m_3264 + 1e-4 and m_h64 <= max(worst_factor * m_3264 + 1e-5, 1e-4), (what, k, "(c)", maxn, m_h64, m_3264)
