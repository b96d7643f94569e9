"""-m gpu: HIP vs the ORACLE at the BASELINE.json sizes, and direct oracle comparisons of the SURVEY §8f entry points.

Round-1 verdict items: (a) no test compared HIP with the oracle above C1 size; (b) the multi-view / abs-grad / folded-loss
entry points were only tested against another HIP path; (c) gradient checks used a max-norm-relative error with a
"10 x the f32 oracle's error" hatch.  Here:

* every BASELINE configuration (C2 200 k, C3 stand-in 262 144 + 81 600 at 512x512 SH1 in the `cube` and `shell`
  layouts, C4 per-GPU share 2 M, C5 500 k surfels; all 800x800 unless said) is compared with the oracle: integer
  state (radii, rects, tiles touched, sorted keys / values, ranges, clamp mask) and per-Gaussian floats bit-exact over
  the WHOLE scene; images, n_contrib and the full backward on a SAMPLE of tiles (the longest lists + random ones —
  oracle tile selection, oracle/oracle_common.h; the backward of a sample equals the full backward with the upstream
  pixel gradients zeroed outside the sample, which is what the HIP path is given);
* gradients are compared PER ELEMENT, |hip - ref| <= 1e-4 |ref| + 1e-6 max|ref| (util.elem_stats), against the
  FLOAT32 oracle — the restatement of the fp32 program the reference runs (its CUDA extension computes in fp32; the
  oracle's K1 stage is bit-identical to ours, its sums are sequential where the GPU's are atomic) — with the fraction
  of elements outside bounded at each assert, no "10 x e_f32" hatch.  The float64 oracle is the arbiter printed next to
  it: measured at these sizes the fp32 ALGORITHM itself is up to 3e-3 (max-norm) from float64 for sub-pixel Gaussians
  (conic = inverse of a nearly singular 2x2 covariance), and HIP and the f32 oracle sit at the same distance from it;
  the tests assert that HIP is no further from float64 than the f32 oracle is (x 1.25);
* render_views (V > 1, activations in K1/K9), render_views_loss (loss in K6/K7) and screenspace_absgrad (+ device top-k)
  are compared with torch autograd through the oracle stand-in (oracle.gdr_oracle.make_standin_module), f32 for the
  bar and f64 as the arbiter.

Reference call sites: /root/reference/lightning/network.py:826-838 (per-view loop), :843-893 (vjp + top-k),
/root/reference/lightning/renderer.py:225-268 (activations, clamp), /root/reference/configs/base.yaml:30 (12 000).
"""
import math
import os

import numpy as np
import pytest
import torch

import util as U

pytestmark = pytest.mark.gpu

THREADS = max(1, min(os.cpu_count() or 1, 64))
GRAD_KEYS = ("means3D", "means2D", "shs", "opacities", "scales", "rotations")


def _case_from_scene(sc, cam, H, W, deg, surfel=False):
    case = dict(
        N=sc["centers"].shape[0], H=H, W=W, deg=deg, means3D=sc["centers"].contiguous(),
        opacities=torch.sigmoid(sc["opacity"]).contiguous(), shs=sc["shs"].contiguous(), colors_precomp=None,
        scales=torch.exp(sc["scales"][:, :2] if surfel else sc["scales"]).contiguous(),
        rotations=torch.nn.functional.normalize(sc["rotations"]).contiguous(), cov3D_precomp=None,
        view=cam.world_view_transform.contiguous(), proj=cam.full_proj_transform.contiguous(),
        campos=cam.camera_center.contiguous(), bg=torch.ones(3), tanfovx=math.tan(0.375), tanfovy=math.tan(0.375),
        scale_modifier=1.0)
    if surfel:
        case["transMat_precomp"] = None
    return case


def _scene(name):
    from generativedensification_amd.camera import orbit_cameras
    from generativedensification_amd.synthetic import make_scene
    if name == "c2":
        sc, H, W, deg, cam_i = make_scene(200_000, 1, sh_degree=3, sigma0=(0.0052, 0.00065)), 800, 800, 3, 1
    elif name == "c4":
        sc, H, W, deg, cam_i = make_scene(2_000_000, 3, sh_degree=3, sigma0=(0.00065,)), 800, 800, 3, 2
    elif name in ("c3cube", "c3shell"):
        lay = name[2:]
        a = make_scene(262_144, 2, sh_degree=1, sigma0=(0.0052,), layout=lay)
        b = make_scene(81_600, 3, sh_degree=1, sigma0=(0.00065,), layout=lay)
        sc, H, W, deg, cam_i = {k: torch.cat([a[k], b[k]]).contiguous() for k in a}, 512, 512, 1, 5
    elif name == "c5":
        sc, H, W, deg, cam_i = make_scene(500_000, 5, sh_degree=3, sigma0=(0.0052, 0.00065)), 800, 800, 3, 3
    else:
        raise KeyError(name)
    cam = orbit_cameras(8, W, H)[cam_i]
    return sc, cam, H, W, deg


_assert_grads = U.assert_grads


@pytest.mark.parametrize("name", ["c2", "c3cube", "c3shell", "c4"])
def test_hip_vs_oracle_at_baseline_sizes(oracle_built, name):
    from oracle.gdr_oracle import Oracle
    sc, cam, H, W, deg = _scene(name)
    case = _case_from_scene(sc, cam, H, W, deg)
    kw = dict(shs=U._np(case["shs"]), scales=U._np(case["scales"]), rotations=U._np(case["rotations"]))
    # HIP forward (whole image) first: its ranges pick the sample
    h, _ = U.run_hip(case)
    tiles = U.pick_tiles(h["ranges"], 16, 32, seed=1)
    mask = U.tile_mask(tiles, H, W)
    o = Oracle("f32", nthreads=THREADS).forward(U._np(case["means3D"]), U._np(case["opacities"]), U.settings_np(case),
                                                tiles=tiles, **kw)
    if name == "c3shell":   # few busy tiles with cut lists: rendered by render_fwd_deep_kernel (16 pixels per wave)
        assert int(h["seg_count"][2]) == 1 and int(h["seg_count"][0]) > 0
    if name in ("c2", "c4"):
        assert int(h["seg_count"][2]) == 0
    # ---- whole scene: integers and per-Gaussian floats bit-exact ---------------------------------------------------
    assert h["num_rendered"] == o["num_rendered"] and o["num_rendered"] > 0
    np.testing.assert_array_equal(h["radii"], o["radii"])
    np.testing.assert_array_equal(h["rect"], o["rect"])
    np.testing.assert_array_equal(h["tiles_touched"].astype(np.uint32), o["tiles_touched"])
    np.testing.assert_array_equal(h["keys_sorted"].view(np.uint64), o["keys_sorted"])
    np.testing.assert_array_equal(h["point_list"].view(np.uint32), o["point_list"])
    np.testing.assert_array_equal(h["ranges"].view(np.uint32), o["ranges"])
    np.testing.assert_array_equal(np.stack([(h["clamped"] >> k) & 1 for k in range(3)], 1), o["clamped"])
    for k in ("depths", "xy", "conic_opacity", "cov3D"):
        np.testing.assert_array_equal(h[k], o[k], err_msg=k)
    np.testing.assert_array_equal(h["rgb"][:, :3], o["rgb"])
    # ---- sampled tiles: images, contributor counts ------------------------------------------------------------------
    nc, ft, px, p = U.assert_image_parity(h, o, name + " sampled tiles", mask)       # counts, not fractions (tests/util.py)
    lens = (o["ranges"][tiles, 1].astype(np.int64) - o["ranges"][tiles, 0])
    print(f"[{name}] D = {o['num_rendered']}, {tiles.size} sampled tiles, list lengths {lens.min()}..{lens.max()}")
    # ---- sampled tiles: full backward (upstream gradients zero outside the sample) ---------------------------------
    grads = [g * torch.from_numpy(mask) for g in U.rand_grads(case)]
    _, hg = U.run_hip(case, grads)
    o64 = Oracle("f64", nthreads=THREADS)
    f64 = o64.forward(U._np(case["means3D"]), U._np(case["opacities"]), U.settings_np(case), tiles=tiles, **kw)
    g64 = o64.backward(f64, *[U._np(g) for g in grads])
    g32 = Oracle("f32", nthreads=1).backward(o, *[U._np(g) for g in grads])   # one thread: sequential f32 sums
    _assert_grads(hg, g64, g32, GRAD_KEYS, name)
    touched = np.zeros(case["N"], bool)
    for t in tiles:
        touched[o["point_list"][o["ranges"][t, 0]:o["ranges"][t, 1]]] = True
    for k in GRAD_KEYS:   # Gaussians outside every sampled tile: exact zeros
        assert not hg[k][~touched].any(), k


def test_whole_images_of_all_bench_views_at_c4_size(oracle_built):
    """Round-2 verdict: at full size the images were compared on 48 of 2500 tiles, the whole-image check was the bench's
    PSNR print.  Here the WHOLE 800x800 image of every view of the C4 bench step (2 M Gaussians, the four cameras bench.py
    renders) is compared with the f32 oracle: colour / depth / alpha per pixel, PSNR, contributor counts and final
    transmittance of every pixel.  (Forward only: the full backward at this size is covered per tile above.)"""
    from oracle.gdr_oracle import Oracle
    from generativedensification_amd.camera import orbit_cameras
    from generativedensification_amd.synthetic import make_scene
    sc = make_scene(2_000_000, 3, sh_degree=3, sigma0=(0.00065,))
    H = W = 800
    cams = orbit_cameras(4, W, H)
    views = range(4) if (os.cpu_count() or 1) >= 16 else range(1)   # (the oracle needs ~4 s per view on 64 threads)
    print(f"[c4 whole image] host has {os.cpu_count()} cpus: comparing view(s) {list(views)} of 4")
    ora = Oracle("f32", nthreads=THREADS)
    for v in views:
        case = _case_from_scene(sc, cams[v], H, W, 3)
        kw = dict(shs=U._np(case["shs"]), scales=U._np(case["scales"]), rotations=U._np(case["rotations"]))
        h, _ = U.run_hip(case)
        o = ora.forward(U._np(case["means3D"]), U._np(case["opacities"]), U.settings_np(case), **kw)
        assert h["num_rendered"] == o["num_rendered"] > 3_000_000
        nc, ft, px, p = U.assert_image_parity(h, o, f"c4 whole image, view {v}")
        print(f"[c4 whole image] view {v}: D = {o['num_rendered']}, PSNR {p:.1f} dB, n_contrib differs on {nc} pixels, "
              f"final T outside on {ft}, colour / depth / alpha outside on {px} (bars {U.MAX_NCONTRIB_PIXELS} / "
              f"{U.MAX_FINAL_T_PIXELS} / {U.MAX_IMAGE_PIXELS} of {H * W})")


def test_whole_image_backward_at_c4_size(oracle_built):
    """... and the backward of a WHOLE image at that size (one view, random upstream gradients on every pixel): all 2 M x 59
    gradient elements per element against the f32 oracle (its per-block partial sums taken by THREADS threads: the sums of
    a Gaussian's pixel terms then associate as on the GPU), float64 as the arbiter."""
    from oracle.gdr_oracle import Oracle
    if (os.cpu_count() or 1) < 16:
        print(f"[c4 whole image backward] SKIPPED: host has {os.cpu_count()} cpus (< 16)")
        pytest.skip("the whole-image oracle backward at 2 M Gaussians needs a many-core host")
    print(f"[c4 whole image backward] host has {os.cpu_count()} cpus: running")
    sc, cam, H, W, deg = _scene("c4")
    case = _case_from_scene(sc, cam, H, W, deg)
    kw = dict(shs=U._np(case["shs"]), scales=U._np(case["scales"]), rotations=U._np(case["rotations"]))
    grads = U.rand_grads(case)
    _, hg = U.run_hip(case, grads)
    out = {}
    for dt in ("f32", "f64"):
        ora = Oracle(dt, nthreads=THREADS)
        f = ora.forward(U._np(case["means3D"]), U._np(case["opacities"]), U.settings_np(case), **kw)
        out[dt] = ora.backward(f, *[U._np(g) for g in grads])
    # max-norm bar 1e-3 instead of 1e-4, per-element bar unchanged: with a gradient on EVERY pixel the float32 algorithm
    # itself sits 3e-3 .. 6e-2 (max-norm) / 0.4 .. 4 % of the elements from float64 on this scene (sub-pixel Gaussians: the conic is the
    # inverse of a nearly singular 2x2 covariance) — HIP and the f32 oracle at the very same distance (asserted, x 1.25) —
    # and for a handful of such Gaussians the ORDER of the pixel sums shows at the 1e-4 .. 7e-4 level (measured: 6e-6 .. 2.2e-5
    # of the elements outside the per-element bar, max-norm 1.7e-4 .. 7.0e-4).
    _assert_grads(hg, out["f64"], out["f32"], GRAD_KEYS, "c4 whole image", maxnorm=1e-3)


@pytest.mark.parametrize("name", ["c2", "c3cube", "c3shell"])
def test_whole_image_forward_and_backward_at_reference_sizes(oracle_built, name):
    """The same on the WHOLE image at the reference-scale configurations (C2, the C3 stand-in in both layouts): every pixel
    of colour / depth / alpha / contributor count / final T, and every gradient element with a random upstream gradient on
    every pixel."""
    from oracle.gdr_oracle import Oracle
    if (os.cpu_count() or 1) < 8:
        pytest.skip("whole-image oracle runs need a multi-core host")
    sc, cam, H, W, deg = _scene(name)
    case = _case_from_scene(sc, cam, H, W, deg)
    kw = dict(shs=U._np(case["shs"]), scales=U._np(case["scales"]), rotations=U._np(case["rotations"]))
    grads = U.rand_grads(case)
    h, hg = U.run_hip(case, grads)
    out = {}
    for dt in ("f32", "f64"):
        ora = Oracle(dt, nthreads=THREADS)
        f = ora.forward(U._np(case["means3D"]), U._np(case["opacities"]), U.settings_np(case), **kw)
        out[dt] = ora.backward(f, *[U._np(g) for g in grads])
        if dt == "f32":
            o = f
    nc, ft, px, p = U.assert_image_parity(h, o, name + " whole image")
    print(f"[{name} whole image] D = {o['num_rendered']}, PSNR {p:.1f} dB, n_contrib differs on {nc} pixels, final T outside on {ft}, "
          f"colour / depth / alpha outside on {px}")
    # (max-norm bar: see the C4 test above.  Fraction outside the per-element bar: measured 0 .. 9.6e-5 here — a random
    # gradient on EVERY pixel makes every Gaussian's sums cancel, which the sampled-tile tests' 48 tiles do not — bar 3e-4.)
    _assert_grads(hg, out["f64"], out["f32"], GRAD_KEYS, name + " whole image", maxnorm=1e-3, max_outside=3e-4)


def test_surfel_hip_vs_oracle_at_c5_size(oracle_built):
    from oracle.gsr_oracle import SurfelOracle
    sc, cam, H, W, deg = _scene("c5")
    case = _case_from_scene(sc, cam, H, W, deg, surfel=True)
    kw = dict(shs=U._np(case["shs"]), scales=U._np(case["scales"]), rotations=U._np(case["rotations"]))
    h, _ = U.run_surfel_hip(case)
    tiles = U.pick_tiles(h["ranges"], 16, 32, seed=2)
    mask = U.tile_mask(tiles, H, W)
    o = SurfelOracle("f32", nthreads=THREADS).forward(U._np(case["means3D"]), U._np(case["opacities"]), U.settings_np(case),
                                                      tiles=tiles, **kw)
    assert h["num_rendered"] == o["num_rendered"] and o["num_rendered"] > 0
    np.testing.assert_array_equal(h["radii"], o["radii"])
    np.testing.assert_array_equal(h["tiles_touched"].astype(np.uint32), o["tiles_touched"])
    np.testing.assert_array_equal(h["keys_sorted"].view(np.uint64), o["keys_sorted"])
    np.testing.assert_array_equal(h["point_list"].view(np.uint32), o["point_list"])
    np.testing.assert_array_equal(h["ranges"].view(np.uint32), o["ranges"])
    assert U.outlier_fraction(h["color"][:, mask], o["color"][:, mask], rtol=1e-4, atol=1e-5) < 1e-4
    for c in range(6):   # expected depth, alpha, normal x3, median depth (distortion: see test_gpu_surfel.py)
        assert U.outlier_fraction(h["allmap"][c][mask], o["allmap"][c][mask], rtol=1e-4, atol=1e-4) < 2e-4, c
    assert U.psnr(np.clip(h["color"][:, mask], 0, 1), np.clip(o["color"][:, mask], 0, 1)) > 60.0
    g = torch.Generator().manual_seed(9)
    grads = [torch.randn(3, H, W, generator=g) * torch.from_numpy(mask), torch.randn(7, H, W, generator=g) * torch.from_numpy(mask)]
    # (the distortion channel's upstream gradient is NOT zeroed any more: the reference weights that channel by 1000,
    # /root/reference/lightning/loss.py:50-53)
    _, hg = U.run_surfel_hip(case, grads)
    o64 = SurfelOracle("f64", nthreads=THREADS)
    f64 = o64.forward(U._np(case["means3D"]), U._np(case["opacities"]), U.settings_np(case), tiles=tiles, **kw)
    g64 = o64.backward(f64, *[U._np(x) for x in grads])
    g32 = SurfelOracle("f32", nthreads=1).backward(o, *[U._np(x) for x in grads])
    # per element against the f32 oracle at the single-call floor (reason + measurements at util.assert_grads_surfel)
    U.assert_grads_surfel(hg, g64, g32, GRAD_KEYS, "c5")


def test_surfel_whole_image_at_c5_size(oracle_built):
    """The 2DGS path on the WHOLE 800x800 image at C5 size (500 k surfels): colour and the six geometric maps of every pixel,
    and every gradient element with a random upstream gradient on every pixel and channel (distortion included)."""
    from oracle.gsr_oracle import SurfelOracle
    if (os.cpu_count() or 1) < 8:
        pytest.skip("whole-image oracle runs need a multi-core host")
    sc, cam, H, W, deg = _scene("c5")
    case = _case_from_scene(sc, cam, H, W, deg, surfel=True)
    kw = dict(shs=U._np(case["shs"]), scales=U._np(case["scales"]), rotations=U._np(case["rotations"]))
    g = torch.Generator().manual_seed(11)
    grads = [torch.randn(3, H, W, generator=g), torch.randn(7, H, W, generator=g)]
    h, hg = U.run_surfel_hip(case, grads)
    out = {}
    for dt in ("f32", "f64"):
        ora = SurfelOracle(dt, nthreads=THREADS)
        f = ora.forward(U._np(case["means3D"]), U._np(case["opacities"]), U.settings_np(case), **kw)
        out[dt] = ora.backward(f, *[U._np(x) for x in grads])
        if dt == "f32":
            o = f
    assert h["num_rendered"] == o["num_rendered"] > 1_000_000
    assert U.outlier_fraction(h["color"], o["color"], rtol=1e-4, atol=1e-5) < 1e-4
    for c in range(6):
        assert U.outlier_fraction(h["allmap"][c], o["allmap"][c], rtol=1e-4, atol=1e-4) < 2e-4, c
    p = U.psnr(np.clip(h["color"], 0, 1), np.clip(o["color"], 0, 1))
    print(f"[c5 whole image] D = {o['num_rendered']}, PSNR {p:.1f} dB")
    assert p > 60.0
    # (a), (b), (c) as everywhere, at the single-call floor (atol_rel 3e-6, 1e-4 of the elements).  Round 4: the intersection
    # runs in the oracle's operation order, so the single worst element of 0.5 M x 59 — one ill-conditioned surfel, which in
    # rounds 1-3 was a different one for every fp32 evaluation order and needed worst_factor 3 — is the oracle's own:
    # measured outside <= 3.5e-6, max-norm vs the f32 oracle <= 2.5e-5, hip-f64 / oracle-f64 = 1.00 (profiles/r04_surfel_stats.txt)
    U.assert_grads_surfel(hg, out["f64"], out["f32"], GRAD_KEYS, "c5 whole image", worst_factor=1.25)


# --------------------------------------------------------------------------------------------------------------------
# §8f entry points against torch float64 autograd through the f64 oracle
# --------------------------------------------------------------------------------------------------------------------
N_MV, H_MV, W_MV, DEG_MV = 30_000, 160, 208, 2


def _mv_setup(V, seed=77):
    from generativedensification_amd.camera import orbit_cameras
    from generativedensification_amd.synthetic import make_scene, make_targets
    sc = make_scene(N_MV, seed, sh_degree=DEG_MV, sigma0=(0.0052, 0.00065, 0.02))
    cams = orbit_cameras(V, W_MV, H_MV)
    tg = make_targets(V, H_MV, W_MV, seed)
    three = ([1.0, 1.0, 1.0], [0.5, 0.5, 0.5], [0.0, 0.0, 0.0])   # gobjverse.py:112-117
    bgs = [torch.tensor(three[j % 3]) for j in range(V)]
    return sc, cams, tg, bgs


def _oracle_views(sc, cams, bgs, loss_of_views, precision="f32", H=None, W=None, deg=None, f32_threads=1):
    """Oracle reference: the reference adaptor's sequence (renderer.py:225-268: sigmoid / exp / normalize, (N,4)
    carrier, rasterizer, clamp) on the ORACLE stand-in, one call per view; gradients by torch autograd."""
    from oracle.gdr_oracle import make_standin_module
    H, W, deg = H or H_MV, W or W_MV, DEG_MV if deg is None else deg
    mod = make_standin_module(precision, nthreads=f32_threads if precision == "f32" else THREADS)
    dt = torch.float32 if precision == "f32" else torch.float64
    leaves = {k: v.to(dt).clone().requires_grad_(True) for k, v in sc.items()}
    ssp = torch.zeros(sc["centers"].shape[0], 4, dtype=dt, requires_grad=True)
    outs = []
    for cam, bg in zip(cams, bgs):
        rs = mod.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(0.375), tanfovy=math.tan(0.375), bg=bg.to(dt),
            scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dt), projmatrix=cam.full_proj_transform.to(dt),
            sh_degree=deg, campos=cam.camera_center.to(dt), prefiltered=False, debug=False)
        color, radii, depth, alpha = mod.GaussianRasterizer(rs)(
            means3D=leaves["centers"], means2D=ssp, shs=leaves["shs"], opacities=torch.sigmoid(leaves["opacity"]),
            scales=torch.exp(leaves["scales"]), rotations=torch.nn.functional.normalize(leaves["rotations"]))
        outs.append(dict(image=color.clamp(0, 1).permute(1, 2, 0), depth=depth.permute(1, 2, 0), acc_map=alpha.squeeze(0)))
    losses = loss_of_views(outs, dt)
    grads = torch.autograd.grad(losses.sum(), list(leaves.values()) + [ssp])
    return losses.detach().numpy(), {k: g.numpy() for k, g in zip(list(leaves) + ["ssp"], grads)}, outs


def _both_oracles(sc, cams, bgs, loss_of_views):
    l32, g32, o32 = _oracle_views(sc, cams, bgs, loss_of_views, "f32")
    l64, g64, _ = _oracle_views(sc, cams, bgs, loss_of_views, "f64")
    return l32, g32, o32, g64


@pytest.mark.parametrize("V", [3, 9])   # 9 > GDR_MAX_VIEWS: two K1/K9 launch groups, the second accumulating
def test_render_views_backward_vs_oracle(oracle_built, V):
    from generativedensification_amd.renderer import Renderer
    from generativedensification_amd.synthetic import view_loss
    dev = torch.device("cuda:0")
    sc, cams, tg, bgs = _mv_setup(V)
    l_ref, g_ref, o_ref, g64 = _both_oracles(sc, cams, bgs, lambda outs, dt: torch.stack(
        [view_loss(o, tg[j].to(dt)) for j, o in enumerate(outs)]))
    r = Renderer(sh_degree=DEG_MV)
    leaves = {k: v.to(dev).clone().requires_grad_(True) for k, v in sc.items()}
    ssp = torch.zeros(N_MV, 4, device=dev, requires_grad=True)
    cams_d = _cams_to(cams, dev)
    outs = r.render_views(cams_d, [b.to(dev) for b in bgs], leaves["centers"], leaves["shs"], leaves["opacity"],
                          leaves["scales"], leaves["rotations"], dev, screenspace_points=ssp)
    for a, b in zip(outs, o_ref):
        for k in ("image", "depth", "acc_map"):
            assert U.outlier_fraction(a[k].detach().cpu().numpy(), b[k].detach().numpy(), 1e-4, 1e-5) < 1e-4, k
    losses = torch.stack([view_loss(o, tg[j].to(dev)) for j, o in enumerate(outs)])
    np.testing.assert_allclose(losses.detach().cpu().numpy(), l_ref, rtol=2e-5)
    grads = torch.autograd.grad(losses.sum(), list(leaves.values()) + [ssp])
    g_hip = {k: g.cpu().numpy() for k, g in zip(list(leaves) + ["ssp"], grads)}
    _assert_grads(g_hip, g64, g_ref, list(g_ref), f"render_views V={V}")


def _cams_to(cams, dev):
    import copy
    out = []
    for c in cams:
        c = copy.copy(c)
        for a in ("world_view_transform", "full_proj_transform", "camera_center"):
            setattr(c, a, getattr(c, a).to(dev))
        out.append(c)
    return out


def test_render_views_loss_vs_oracle(oracle_built):
    """The loss folded into K6's epilogue / K7's prologue (gdr_composite_forward_loss / gdr_render_backward_loss)."""
    from generativedensification_amd.renderer import Renderer
    from generativedensification_amd.synthetic import view_loss
    dev = torch.device("cuda:0")
    V = 4
    sc, cams, tg, bgs = _mv_setup(V, seed=78)
    l_ref, g_ref, _, g64 = _both_oracles(sc, cams, bgs, lambda outs, dt: torch.stack(
        [view_loss(o, tg[j].to(dt)) for j, o in enumerate(outs)]))
    r = Renderer(sh_degree=DEG_MV)
    leaves = {k: v.to(dev).clone().requires_grad_(True) for k, v in sc.items()}
    ssp = torch.zeros(N_MV, 4, device=dev, requires_grad=True)
    lv = r.render_views_loss(_cams_to(cams, dev), [b.to(dev) for b in bgs], tg.permute(0, 3, 1, 2).contiguous().to(dev),
                             leaves["centers"], leaves["shs"], leaves["opacity"], leaves["scales"], leaves["rotations"], dev,
                             screenspace_points=ssp)
    np.testing.assert_allclose(lv.detach().cpu().numpy(), l_ref, rtol=2e-5)
    grads = torch.autograd.grad(lv.sum(), list(leaves.values()) + [ssp])
    g_hip = {k: g.cpu().numpy() for k, g in zip(list(leaves) + ["ssp"], grads)}
    _assert_grads(g_hip, g64, g_ref, list(g_ref), "render_views_loss")


def test_bench_step_at_c4_size_vs_oracle(oracle_built):
    """The step bench.py times by default — 2 M Gaussians, four 800x800 views in ONE fused node with the loss folded into
    K6 / K7 (Renderer.render_views_loss), one backward — against the oracle doing the same through the reference adaptor's
    op sequence, one `GaussianRasterizer` call per view, torch autograd, float32 (the bar) and float64 (the arbiter): the
    four per-view losses and every element of the six leaf gradients + the (N,4) carrier's."""
    from generativedensification_amd.camera import orbit_cameras
    from generativedensification_amd.renderer import Renderer
    from generativedensification_amd.synthetic import make_scene, make_targets, view_loss
    if (os.cpu_count() or 1) < 16:
        pytest.skip("eight oracle passes over 2 M Gaussians need a many-core host")
    dev = torch.device("cuda:0")
    N, H, W, deg, V = 2_000_000, 800, 800, 3, 4
    sc = make_scene(N, 3, sh_degree=deg, sigma0=(0.00065,))
    cams = orbit_cameras(V, W, H)
    tg = make_targets(V, H, W, 3)
    bgs = [torch.ones(3) for _ in range(V)]
    loss_fn = lambda outs, dt: torch.stack([view_loss(o, tg[j].to(dt)) for j, o in enumerate(outs)])
    l32, g32, _ = _oracle_views(sc, cams, bgs, loss_fn, "f32", H, W, deg, f32_threads=THREADS)
    l64, g64, _ = _oracle_views(sc, cams, bgs, loss_fn, "f64", H, W, deg)
    leaves = {k: v.to(dev).clone().requires_grad_(True) for k, v in sc.items()}
    ssp = torch.zeros(N, 4, device=dev, requires_grad=True)
    r = Renderer(sh_degree=deg)
    lv = r.render_views_loss(_cams_to(cams, dev), [b.to(dev) for b in bgs], tg.permute(0, 3, 1, 2).contiguous().to(dev),
                             leaves["centers"], leaves["shs"], leaves["opacity"], leaves["scales"], leaves["rotations"], dev,
                             screenspace_points=ssp)
    print("[c4 bench step] losses hip", lv.detach().cpu().numpy(), "f32 oracle", l32, "f64", l64)
    np.testing.assert_allclose(lv.detach().cpu().numpy(), l32, rtol=2e-5)
    grads = torch.autograd.grad(lv.sum(), list(leaves.values()) + [ssp])
    g_hip = {k: g.cpu().numpy() for k, g in zip(list(leaves) + ["ssp"], grads)}
    # Per-element bar as in the whole-image tests (every pixel of four views carries gradient).  Max-norm, i.e. the single
    # worst of 2 M x 63 elements: the float32 algorithm itself is 1.8e-3 .. 8e-3 from float64 here in that norm (0.3 .. 3.5 % of the elements outside the per-element bar) (sub-pixel
    # Gaussians, sums over four views), HIP and the f32 oracle at the SAME distance (asserted, x 1.25); between the two fp32
    # evaluations the worst element differs by 2e-4 .. 2.0e-3, 8e-6 .. 4.2e-5 of the elements are outside — bars 3e-3 / 3e-4.
    _assert_grads(g_hip, g64, g32, list(g32), "c4 bench step", maxnorm=3e-3, max_outside=3e-4)
    # ... and the same step through the UNCHANGED caller's loop (bench.py's `per_view`): render_img per view with torch
    # activations and a new carrier per call, torch loss, one backward — the four calls form a render group (viewgroup.py)
    from generativedensification_amd import viewgroup as VG
    r2 = Renderer(sh_degree=deg, fused=False)
    leaves2 = {k: v.to(dev).clone().requires_grad_(True) for k, v in sc.items()}
    cams_d, tg_d = _cams_to(cams, dev), tg.to(dev)
    losses2, carriers = [], []
    for j, cam in enumerate(cams_d):
        r2.set_bg_color(bgs[j].to(dev))
        ssp_j = torch.zeros(N, 4, device=dev, requires_grad=True)
        out = r2.render_img(cam, None, leaves2["centers"], leaves2["shs"], leaves2["opacity"], leaves2["scales"],
                            leaves2["rotations"], dev, screenspace_points=ssp_j)
        losses2.append(view_loss(out, tg_d[j]))
        carriers.append(ssp_j)
    lv2 = torch.stack(losses2)
    assert V in VG.live_group_views(), "the four calls did not form one render group"
    np.testing.assert_allclose(lv2.detach().cpu().numpy(), l32, rtol=2e-5)
    grads2 = torch.autograd.grad(lv2.sum(), list(leaves2.values()) + carriers)
    g_hip2 = {k: g.cpu().numpy() for k, g in zip(list(leaves2), grads2[:len(leaves2)])}
    g_hip2["ssp"] = sum(g.cpu().numpy() for g in grads2[len(leaves2):])     # (the oracle's one carrier = the sum over the views)
    _assert_grads(g_hip2, g64, g32, list(g32), "c4 bench step, unchanged caller", maxnorm=3e-3, max_outside=3e-4)


def test_ten_views_at_c2_size_chunked_launches_vs_oracle(oracle_built):
    """V = 10 > GDR_MAX_VIEWS at a BASELINE size (C2: 200 k Gaussians, 800x800, SH 3; round-3 verdict: the <= 8-views-per-
    launch chunking with the accumulate flag was only tested on small scenes): the fused node (K1 in two launches, K7 of the
    views in two launches, K8+K9 in two launches, the second accumulating) and the unchanged caller's loop (one render group
    of ten calls: the hub's K8+K9 in two launches) against the oracle run through the reference adaptor's op sequence.
    The fused node runs with the row-pair K7 pinned (render_bwd_pairs_kernel, the variant the library settles on for this
    scene), the unchanged caller with the row kernel: both K7 kernels against the oracle at a full size."""
    from generativedensification_amd import _lib as L
    from generativedensification_amd import viewgroup as VG
    from generativedensification_amd.camera import orbit_cameras
    from generativedensification_amd.renderer import Renderer
    from generativedensification_amd.synthetic import make_scene, make_targets, view_loss
    if (os.cpu_count() or 1) < 16:
        pytest.skip("twenty oracle passes at 800x800 need a many-core host")
    dev = torch.device("cuda:0")
    N, H, W, deg, V = 200_000, 800, 800, 3, 10
    sc = make_scene(N, 1, sh_degree=deg, sigma0=(0.0052, 0.00065))
    cams = orbit_cameras(V, W, H)
    tg = make_targets(V, H, W, 1)
    three = ([1.0, 1.0, 1.0], [0.5, 0.5, 0.5], [0.0, 0.0, 0.0])
    bgs = [torch.tensor(three[j % 3]) for j in range(V)]
    loss_fn = lambda outs, dt: torch.stack([view_loss(o, tg[j].to(dt)) for j, o in enumerate(outs)])
    l32, g32, _ = _oracle_views(sc, cams, bgs, loss_fn, "f32", H, W, deg, f32_threads=THREADS)
    l64, g64, _ = _oracle_views(sc, cams, bgs, loss_fn, "f64", H, W, deg)
    cams_d, tg_d = _cams_to(cams, dev), tg.to(dev)
    leaves = {k: v.to(dev).clone().requires_grad_(True) for k, v in sc.items()}
    ssp = torch.zeros(N, 4, device=dev, requires_grad=True)
    r = Renderer(sh_degree=deg)
    lv = r.render_views_loss(cams_d, [b.to(dev) for b in bgs], tg.permute(0, 3, 1, 2).contiguous().to(dev), leaves["centers"],
                             leaves["shs"], leaves["opacity"], leaves["scales"], leaves["rotations"], dev, screenspace_points=ssp)
    np.testing.assert_allclose(lv.detach().cpu().numpy(), l32, rtol=2e-5)
    L.load().gdr_k7_tune_override(1)
    try:
        grads = torch.autograd.grad(lv.sum(), list(leaves.values()) + [ssp])
    finally:
        L.load().gdr_k7_tune_override(0)
    g_hip = {k: g.cpu().numpy() for k, g in zip(list(leaves) + ["ssp"], grads)}
    _assert_grads(g_hip, g64, g32, list(g32), "c2 ten views, fused, row-pair K7", maxnorm=3e-3, max_outside=3e-4)
    VG.pace().solo_passes = 0
    r2 = Renderer(sh_degree=deg, fused=False)
    leaves2 = {k: v.to(dev).clone().requires_grad_(True) for k, v in sc.items()}
    losses2, carriers = [], []
    for j, cam in enumerate(cams_d):
        r2.set_bg_color(bgs[j].to(dev))
        ssp_j = torch.zeros(N, 4, device=dev, requires_grad=True)
        out = r2.render_img(cam, None, leaves2["centers"], leaves2["shs"], leaves2["opacity"], leaves2["scales"],
                            leaves2["rotations"], dev, screenspace_points=ssp_j)
        losses2.append(view_loss(out, tg_d[j]))
        carriers.append(ssp_j)
    lv2 = torch.stack(losses2)
    assert V in VG.live_group_views(), "the ten calls did not form one render group"
    np.testing.assert_allclose(lv2.detach().cpu().numpy(), l32, rtol=2e-5)
    try:
        grads2 = torch.autograd.grad(lv2.sum(), list(leaves2.values()) + carriers)
    finally:
        L.load().gdr_k7_tune_override(-1)
    g_hip2 = {k: g.cpu().numpy() for k, g in zip(list(leaves2), grads2[:len(leaves2)])}
    g_hip2["ssp"] = sum(g.cpu().numpy() for g in grads2[len(leaves2):])
    _assert_grads(g_hip2, g64, g32, list(g32), "c2 ten views, unchanged caller, row K7", maxnorm=3e-3, max_outside=3e-4)


def test_screenspace_absgrad_and_topk_vs_oracle(oracle_built):
    """SURVEY §8f-2 (network.py:843-893): MSE over 4 views differentiated w.r.t. the (N,4) carrier only, then the
    top-k of ||grad[:, 2:4]||.  Reference = sum over the views of the f64 oracle's mean2D gradients."""
    from generativedensification_amd.renderer import Renderer
    dev = torch.device("cuda:0")
    V, k = 4, 3000
    sc, cams, tg, bgs = _mv_setup(V, seed=79)
    l_ref, g_ref, _, g64 = _both_oracles(sc, cams, bgs, lambda outs, dt: (
        (torch.stack([o["image"] for o in outs]) - tg.to(dt)) ** 2).mean().reshape(1))
    loss, grad, idx = Renderer(sh_degree=DEG_MV).screenspace_absgrad(
        _cams_to(cams, dev), [b.to(dev) for b in bgs], tg.to(dev), *[sc[n].to(dev) for n in
                                                                    ("centers", "shs", "opacity", "scales", "rotations")],
        dev, topk=k)
    assert abs(float(loss) - float(l_ref[0])) <= 2e-5 * abs(float(l_ref[0]))
    grad = grad.cpu().numpy()
    assert grad.shape == (N_MV, 4) and (grad[:, 2:] >= 0).all()
    _assert_grads({"ssp": grad}, g64, g_ref, ["ssp"], "absgrad")
    # top-k: exactly the oracle's selection wherever the selection is well separated
    score = np.linalg.norm(g_ref["ssp"][:, 2:4], axis=1)
    order = np.argsort(-score, kind="stable")
    kth = score[order[k - 1]]
    sure_in = set(np.nonzero(score > kth * (1 + 1e-3))[0].tolist())
    sure_out = set(np.nonzero(score < kth * (1 - 1e-3))[0].tolist())
    got = set(idx.cpu().tolist())
    assert len(got) == k and sure_in <= got and not (got & sure_out)


@pytest.mark.parametrize("size", ["small", "c5"])
def test_surfel_render_views_backward_vs_oracle(oracle_built, size):
    """The 2DGS multi-view node (K1s / K9s for all views, activations folded in) against torch autograd through the
    SURFEL ORACLE stand-in, one call per view (renderer_2dgs.py:92-96, 224-234 semantics) — round 1 only compared it
    with the per-view HIP sequence.  Upstream: random gradients on the colour and on all seven allmap channels
    (the distortion channel included: the reference weights it by 1000, loss.py:50-53)."""
    from generativedensification_amd.camera import build_rays, orbit_cameras
    from generativedensification_amd.renderer_2dgs import Renderer
    from generativedensification_amd.synthetic import make_scene
    from oracle.gsr_oracle import make_surfel_standin_module
    dev = torch.device("cuda:0")
    if size == "c5":   # the C5 bench step's node: 500 k surfels, four 800x800 views in one node
        if (os.cpu_count() or 1) < 16:
            pytest.skip("eight surfel-oracle passes over 500 k surfels need a many-core host")
        V, n, h, w, deg, sig, f32_threads = 4, 500_000, 800, 800, 3, (0.0052, 0.00065), THREADS
    else:
        V, n, h, w, deg, sig, f32_threads = 3, 20_000, 144, 176, 2, (0.0052, 0.02), 1
    sc = make_scene(n, 81 if size == "small" else 5, sh_degree=deg, sigma0=sig)
    sc["scales"] = sc["scales"][:, :2].contiguous()
    cams = orbit_cameras(V, w, h)
    g = torch.Generator().manual_seed(5)
    gc = [torch.randn(3, h, w, generator=g) for _ in range(V)]
    ga = [torch.randn(7, h, w, generator=g) for _ in range(V)]

    def oracle_ref(precision):
        mod = make_surfel_standin_module(precision, nthreads=f32_threads if precision == "f32" else THREADS)
        dt = torch.float32 if precision == "f32" else torch.float64
        leaves = {k: v.to(dt).clone().requires_grad_(True) for k, v in sc.items()}
        ssp = torch.zeros(n, 4, dtype=dt, requires_grad=True)
        total, outs = 0, []
        for v, cam in enumerate(cams):
            rs = mod.GaussianRasterizationSettings(
                image_height=h, image_width=w, tanfovx=math.tan(0.375), tanfovy=math.tan(0.375), bg=torch.ones(3, dtype=dt),
                scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dt), projmatrix=cam.full_proj_transform.to(dt),
                sh_degree=deg, campos=cam.camera_center.to(dt), prefiltered=False, debug=False)
            color, radii, allmap = mod.GaussianRasterizer(rs)(
                means3D=leaves["centers"], means2D=ssp, shs=leaves["shs"], opacities=torch.sigmoid(leaves["opacity"]),
                scales=torch.exp(leaves["scales"]), rotations=torch.nn.functional.normalize(leaves["rotations"]))
            total = total + (color * gc[v].to(dt)).sum() + (allmap * ga[v].to(dt)).sum()
            outs.append((color.detach().numpy(), allmap.detach().numpy()))
        grads = torch.autograd.grad(total, list(leaves.values()) + [ssp])
        return outs, {k: x.numpy() for k, x in zip(list(leaves) + ["ssp"], grads)}

    o32, g32 = oracle_ref("f32")
    _, g64 = oracle_ref("f64")
    r = Renderer(sh_degree=deg, white_background=True)
    leaves = {k: v.to(dev).clone().requires_grad_(True) for k, v in sc.items()}
    ssp = torch.zeros(n, 4, device=dev, requires_grad=True)
    cams_d = _cams_to(cams, dev)
    rays = [build_rays(torch.inverse(c.world_view_transform.T.cpu()), 0.75, 0.75, h, w).to(dev) for c in cams]
    outs = r.render_views(cams_d, rays, None, leaves["centers"], leaves["shs"], leaves["opacity"], leaves["scales"],
                          leaves["rotations"], dev, screenspace_points=ssp, raw=True)
    total = 0
    for v, o in enumerate(outs):
        assert U.outlier_fraction(o["color"].detach().cpu().numpy(), o32[v][0], 1e-4, 1e-5) < 1e-4
        for c in range(6):
            assert U.outlier_fraction(o["allmap"][c].detach().cpu().numpy(), o32[v][1][c], 1e-4, 1e-4) < 2e-4, c
        total = total + (o["color"] * gc[v].to(dev)).sum() + (o["allmap"] * ga[v].to(dev)).sum()
    grads = torch.autograd.grad(total, list(leaves.values()) + [ssp])
    g_hip = {k: x.cpu().numpy() for k, x in zip(list(leaves) + ["ssp"], grads)}
    # (the RAW entry: sigmoid / exp / normalize run inside K1s and differ from torch's in the last bit, which the
    # ill-conditioned geometry amplifies — the 1e-5 floor and 2e-4 of the elements outside, util.py; the single worst element
    # is within 1.25 x of the f32 oracle's own distance from float64 since the intersection runs in the oracle's order,
    # round 4: measured ratio 1.00 at both sizes, profiles/r04_surfel_stats.txt)
    U.assert_grads_surfel(g_hip, g64, g32, list(g32), "surfel render_views " + size, worst_factor=1.25,
                          max_outside=U.SURFEL_RAW_MAX_OUTSIDE, atol_rel=U.SURFEL_RAW_ATOL_REL)
    # ... and the same views through the UNCHANGED caller's loop of renderer_2dgs.py:224-234 — torch activations, one
    # `diff_surfel_rasterization.GaussianRasterizer` call per view, one backward: the calls form a render group (round 4), one
    # K9s for all of them; activated inputs, so the single-call bar (3e-6 floor, 1e-4 of the elements) applies
    import diff_surfel_rasterization as DS
    from generativedensification_amd import viewgroup as VG
    VG.pace().solo_passes = 0
    leaves2 = {k: v.to(dev).clone().requires_grad_(True) for k, v in sc.items()}
    ssp2 = [torch.zeros(n, 4, device=dev, requires_grad=True) for _ in range(V)]
    total2 = 0
    for v, cam in enumerate(cams_d):
        rs = DS.GaussianRasterizationSettings(
            image_height=h, image_width=w, tanfovx=math.tan(0.375), tanfovy=math.tan(0.375), bg=torch.ones(3, device=dev),
            scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=deg,
            campos=cam.camera_center, prefiltered=False, debug=False)
        color, radii, allmap = DS.GaussianRasterizer(rs)(
            means3D=leaves2["centers"], means2D=ssp2[v], shs=leaves2["shs"], opacities=torch.sigmoid(leaves2["opacity"]),
            scales=torch.exp(leaves2["scales"]), rotations=torch.nn.functional.normalize(leaves2["rotations"]))
        total2 = total2 + (color * gc[v].to(dev)).sum() + (allmap * ga[v].to(dev)).sum()
    assert V in VG.live_group_views(), "no surfel render group formed"
    grads2 = torch.autograd.grad(total2, list(leaves2.values()) + ssp2)
    g_grp = {k: x.cpu().numpy() for k, x in zip(list(leaves2), grads2[:len(leaves2)])}
    g_grp["ssp"] = sum(x.cpu().numpy() for x in grads2[len(leaves2):])       # (the oracle's one carrier = the sum over the views)
    # (sums over the V views of per-view gradients of mixed sign — the opacity logits' above all — in another order than the
    # oracle's autograd adds them: the multi-view bars, as for the fused node above; measured at C5: <= 1.2e-4 outside at the
    # 3e-6 floor, <= 8e-5 at 1e-5)
    U.assert_grads_surfel(g_grp, g64, g32, list(g32), "surfel group " + size, worst_factor=1.25,
                          max_outside=U.SURFEL_RAW_MAX_OUTSIDE, atol_rel=U.SURFEL_RAW_ATOL_REL)
