"""-m gpu: the reference's REAL per-sample render sequence as one oracle-checked workload (BASELINE configs[2]).

/root/reference/lightning/network.py, one sample of a training step:
  :826-838  8 coarse renders of the 64^3 x K = 262 144 coarse Gaussians (512x512, SH degree 1, per-view background);
  :843-893  `vjp` of the image MSE over the first 4 views w.r.t. a shared (N,4) carrier -> ||grad[:, 2:4]|| -> top-k 12 000
            (configs/base.yaml:30);
  :949-959  fine set = 81 600 new Gaussians + the coarse ones that were not selected;
  :964-972  8 fine renders of that DIFFERENT set;
  loss on image + image_fine (loss.py:37-48), with depth / alpha of the COARSE renders carrying gradient (they feed
  grid_sample un-detached, network.py:746-752) -> ONE backward through all 16 render graphs.

The sequence is written once (`_reference_step`, op for op what renderer.py:225-268 and network.py do) and run
  * on the ORACLE stand-in (f32 = the bar, f64 = the arbiter), CPU;
  * on the product boundary `diff_gaussian_rasterization` exactly as the unchanged caller would (render groups);
  * through the fused entry points (`Renderer.render_views`, `screenspace_absgrad(topk=)`).
Losses, the top-k selection and every leaf gradient of both HIP runs are compared with the oracle's."""
import math
import os

import numpy as np
import pytest
import torch

import util as U

pytestmark = pytest.mark.gpu
THREADS = max(1, min(os.cpu_count() or 1, 64))
H = W = 512
DEG = 1
N_COARSE, N_FINE, K_NUM, V_ALL, V_SEL = 262_144, 81_600, 12_000, 8, 4
KEYS = ("centers", "shs", "opacity", "scales", "rotations")


def _inputs():
    from generativedensification_amd.camera import orbit_cameras
    from generativedensification_amd.synthetic import make_scene, make_targets
    coarse = make_scene(N_COARSE, 2, sh_degree=DEG, sigma0=(0.0052,))
    fine = make_scene(N_FINE, 3, sh_degree=DEG, sigma0=(0.00065,))
    cams = orbit_cameras(V_ALL, W, H)
    tg = make_targets(V_ALL, H, W, 2)
    three = ([1.0, 1.0, 1.0], [0.5, 0.5, 0.5], [0.0, 0.0, 0.0])      # dataLoader/gobjverse.py:112-117
    bgs = [torch.tensor(three[j % 3]) for j in range(V_ALL)]
    return coarse, fine, cams, tg, bgs


def _render_img(mod, cam, bg, centers, shs, opacity, scales, rotations, dt, dev, ssp=None):
    """lightning/renderer.py:209-272, op for op (set_rasterizer :106-126 included)."""
    rs = mod.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(0.375), tanfovy=math.tan(0.375), bg=bg.to(dev, dt),
        scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dev, dt), projmatrix=cam.full_proj_transform.to(dev, dt),
        sh_degree=DEG, campos=cam.camera_center.to(dev, dt), prefiltered=False, debug=False)
    opacity = torch.sigmoid(opacity)
    scales = torch.exp(scales)
    rotations = torch.nn.functional.normalize(rotations)
    if ssp is None:
        ssp = torch.zeros((centers.shape[0], 4), dtype=centers.dtype, requires_grad=True, device=dev) + 0
    try:
        ssp.retain_grad()
    except Exception:
        pass
    image, radii, depth, alpha = mod.GaussianRasterizer(raster_settings=rs)(
        means3D=centers, means2D=ssp, shs=shs, opacities=opacity, scales=scales, rotations=rotations, cov3D_precomp=None)
    return {"image": image.clamp(0, 1).permute(1, 2, 0), "depth": depth.permute(1, 2, 0), "acc_map": alpha.squeeze(0)}


def _reference_step(mod, dt, dev, coarse, fine, cams, tg, bgs, force_mask=None):
    from torch.autograd.functional import vjp
    lc = {k: v.to(dev, dt).clone().requires_grad_(True) for k, v in coarse.items()}
    lf = {k: v.to(dev, dt).clone().requires_grad_(True) for k, v in fine.items()}
    tgd = tg.to(dev, dt)
    a = [lc[k] for k in KEYS]
    outs_c = [_render_img(mod, cams[j], bgs[j], *a, dt, dev) for j in range(V_ALL)]                # network.py:826-838

    def fn(ssp):                                                                                    # network.py:843-863
        frames = [_render_img(mod, cams[j], bgs[j], *a, dt, dev, ssp=ssp) for j in range(V_SEL)]
        return ((torch.stack([f["image"] for f in frames]) - tgd[:V_SEL]) ** 2).mean()
    image_loss, grad = vjp(fn, torch.zeros(N_COARSE, 4, dtype=dt, device=dev))                     # network.py:867-872
    score = torch.norm(grad[:, 2:4], dim=-1)                                                        # network.py:878
    own = torch.zeros(N_COARSE, dtype=torch.bool, device=dev)
    own[torch.topk(score, K_NUM, dim=0).indices] = True                                             # network.py:888-890
    mask = own if force_mask is None else force_mask.to(dev)
    fs = {k: torch.cat([lf[k], lc[k][~mask]], dim=0) for k in KEYS}                                 # network.py:949-959
    outs_f = [_render_img(mod, cams[j], bgs[j], *[fs[k] for k in KEYS], dt, dev) for j in range(V_ALL)]   # :964-972
    per_view = torch.stack([((oc["image"] - tgd[j]) ** 2).mean() + ((of["image"] - tgd[j]) ** 2).mean()
                            + 0.1 * oc["depth"].mean() + 0.1 * oc["acc_map"].mean()
                            for j, (oc, of) in enumerate(zip(outs_c, outs_f))])
    per_view.sum().backward()                                                                       # ONE backward
    g = {f"coarse_{k}": lc[k].grad.detach().cpu().numpy() for k in KEYS}
    g.update({f"fine_{k}": lf[k].grad.detach().cpu().numpy() for k in KEYS})
    return dict(per_view=per_view.detach().cpu().numpy(), image_loss=float(image_loss), grad=grad.detach().cpu().numpy(),
                score=score.detach().cpu().numpy(), own_mask=own.cpu(), g=g)


def _fused_step(dev, coarse, fine, cams, tg, bgs, force_mask):
    """The same sample through the fused entry points a modified caller can use."""
    from generativedensification_amd.renderer import Renderer
    r = Renderer(sh_degree=DEG)
    lc = {k: v.to(dev).clone().requires_grad_(True) for k, v in coarse.items()}
    lf = {k: v.to(dev).clone().requires_grad_(True) for k, v in fine.items()}
    tgd = tg.to(dev)
    cd, bd = _cams_to(cams, dev), [b.to(dev) for b in bgs]
    a = [lc[k] for k in KEYS]
    outs_c = r.render_views(cd, bd, *a, dev)
    image_loss, grad, idx = r.screenspace_absgrad(cd[:V_SEL], bd[:V_SEL], tgd[:V_SEL], *[x.detach() for x in a], dev, topk=K_NUM)
    own = torch.zeros(N_COARSE, dtype=torch.bool, device=dev)
    own[idx] = True
    mask = force_mask.to(dev)
    fs = {k: torch.cat([lf[k], lc[k][~mask]], dim=0) for k in KEYS}
    outs_f = r.render_views(cd, bd, *[fs[k] for k in KEYS], dev)
    per_view = torch.stack([((oc["image"] - tgd[j]) ** 2).mean() + ((of["image"] - tgd[j]) ** 2).mean()
                            + 0.1 * oc["depth"].mean() + 0.1 * oc["acc_map"].mean()
                            for j, (oc, of) in enumerate(zip(outs_c, outs_f))])
    per_view.sum().backward()
    g = {f"coarse_{k}": lc[k].grad.detach().cpu().numpy() for k in KEYS}
    g.update({f"fine_{k}": lf[k].grad.detach().cpu().numpy() for k in KEYS})
    return dict(per_view=per_view.detach().cpu().numpy(), image_loss=float(image_loss), grad=grad.cpu().numpy(),
                score=torch.norm(grad[:, 2:4], dim=-1).cpu().numpy(), own_mask=own.cpu(), g=g)


def _cams_to(cams, dev):
    import copy
    out = []
    for c in cams:
        c = copy.copy(c)
        for a in ("world_view_transform", "full_proj_transform", "camera_center"):
            setattr(c, a, getattr(c, a).to(dev))
        out.append(c)
    return out


def test_reference_training_sample_sequence_vs_oracle(oracle_built):
    import diff_gaussian_rasterization as D
    from generativedensification_amd import _lib as L
    from oracle.gdr_oracle import make_standin_module
    coarse, fine, cams, tg, bgs = _inputs()
    cpu = torch.device("cpu")
    # (f32 oracle on all host cores: its per-Gaussian sums are per-4x4-block partial sums added with `omp atomic` — the
    # order of those few adds varies at the 1e-7 level, far below the bar; one thread would take minutes at this size)
    o32 = _reference_step(make_standin_module("f32", nthreads=THREADS), torch.float32, cpu, coarse, fine, cams, tg, bgs)
    mask = o32["own_mask"]
    o64 = _reference_step(make_standin_module("f64", nthreads=THREADS), torch.float64, cpu, coarse, fine, cams, tg, bgs,
                          force_mask=mask)
    dev = torch.device("cuda:0")
    L.profile_enable(True)
    L.profile_collect(reset=True)
    hip = _reference_step(D, torch.float32, dev, coarse, fine, _cams_to(cams, dev), tg, bgs, force_mask=mask)
    torch.cuda.synchronize()
    prof = L.profile_collect(reset=True)
    L.profile_enable(False)
    # the unchanged caller: 8 + 4 + 8 render calls; K7 for each of them; but (render groups) ONE preprocess-backward per
    # Gaussian set — the vjp pass runs K7 only (its mean2D-only form: the pass never reaches the group's hub), the main pass one
    # multi-view K8+K9 for the coarse and one for the fine set — and (round 5) NO forward for the 4 views the vjp pass repeats
    # (network.py:848-856 renders the first n_views_sel views of the same Gaussians with the same c2w / bg again): 16 K1 / K6
    # launches, not 20
    assert prof["preprocess_fwd"][1] == 2 * V_ALL and prof["render_fwd"][1] == 2 * V_ALL, (prof["preprocess_fwd"], prof["render_fwd"])
    assert prof["render_bwd"][1] == 2 * V_ALL + V_SEL and prof["preprocess_bwd"][1] == 2, (prof["render_bwd"], prof["preprocess_bwd"])
    fused = _fused_step(dev, coarse, fine, cams, tg, bgs, mask)

    score = o32["score"].astype(np.float64)
    kth = np.sort(score)[-K_NUM]
    sure_in = torch.from_numpy(score > kth * (1 + 1e-3))
    sure_out = torch.from_numpy(score < kth * (1 - 1e-3))
    for name, run in (("unchanged caller", hip), ("fused entries", fused)):
        np.testing.assert_allclose(run["per_view"], o32["per_view"], rtol=2e-5, err_msg=name)
        assert abs(run["image_loss"] - o32["image_loss"]) <= 2e-5 * abs(o32["image_loss"]), name
        U.assert_grads({"ssp": run["grad"]}, {"ssp": o64["grad"]}, {"ssp": o32["grad"]}, ["ssp"], f"{name}: vjp carrier")
        assert (run["grad"][:, 2:] >= 0).all()
        own = run["own_mask"]
        assert int(own.sum()) == K_NUM and bool(own[sure_in].all()) and not bool(own[sure_out].any()), name
        # Leaf gradients: sums over 16 full-size renders (8 coarse incl. depth / alpha terms + 8 fine) of Gaussians that
        # are mostly 2-5 px wide.  Measured: the f32 ORACLE ITSELF is 3.3e-3 of the elements / 1.1e-3 max-norm from
        # float64 here, and every marginal alpha >= 1/255 decision that flips between two fp32 evaluations (counted,
        # < 1e-4 of the pixels, in the per-render tests) moves all Gaussians under that pixel — HIP vs the f32 oracle
        # at most 2.2e-4 of the elements / 3.2e-4 max-norm (profiles/r03_reference_step_parity.txt).  Bars: 5e-4 / 1e-3, still
        # 6 x / 1 x below the fp32 algorithm's own distance from float64; HIP no further from float64 than the f32
        # oracle (asserted inside, x 1.25).
        U.assert_grads(run["g"], o64["g"], o32["g"], list(o32["g"]), name, max_outside=5e-4, maxnorm=1e-3)
    # culled / unselected bookkeeping: the coarse Gaussians that were densified away still get gradient from the coarse renders
    assert np.abs(hip["g"]["coarse_opacity"][mask.numpy()]).max() > 0
