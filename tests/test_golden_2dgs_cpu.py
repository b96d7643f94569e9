"""CPU: committed 2DGS fixtures (tests/golden/render2dgs_*.npz, produced by make_golden_2dgs.py with the REFERENCE's
surfel adaptor /root/reference/lightning/renderer_2dgs.py on top of the surfel oracle) vs
 (a) the repo's host-side mirror generativedensification_amd/renderer_2dgs.py (allmap slicing, normal rotation,
     expected/median depth, depth_to_normal, (N,4) carrier) with the oracle stand-in underneath, and
 (b) the oracle as it builds today (regression)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OUT_KEYS = ("image", "depth", "acc_map", "rend_normal", "depth_normal", "rend_dist")


def _load(name):
    return dict(np.load(os.path.join(GOLD, name)))


def make_cam(g, device="cpu"):
    from generativedensification_amd.camera import MiniCam

    return MiniCam(torch.from_numpy(g["c2w"]), int(g["w"]), int(g["h"]), torch.tensor(float(g["fov"])),
                   torch.tensor(float(g["fov"])), float(g["znear"]), float(g["zfar"]), device)


@pytest.mark.parametrize("name", ["render2dgs_deg3.npz", "render2dgs_deg1_median.npz"])
def test_2dgs_mirror_reproduces_reference_render_img(oracle_built, monkeypatch, name):
    from generativedensification_amd import renderer_2dgs as R
    from generativedensification_amd.camera import build_rays
    from generativedensification_amd.synthetic import surfel_loss
    from oracle.gsr_oracle import make_surfel_standin_module

    g = _load(name)
    st = make_surfel_standin_module("f32")
    monkeypatch.setattr(R, "GaussianRasterizationSettings", st.GaussianRasterizationSettings)
    monkeypatch.setattr(R, "GaussianRasterizer", st.GaussianRasterizer)
    cam = make_cam(g)
    for k in ("world_view_transform", "full_proj_transform", "camera_center"):
        np.testing.assert_array_equal(getattr(cam, k).numpy(), g[k], err_msg=k)
    h, w, n = int(g["h"]), int(g["w"]), int(g["n"])
    rays = build_rays(torch.from_numpy(g["c2w"]), float(g["fov"]), float(g["fov"]), h, w)
    np.testing.assert_array_equal(rays.numpy(), g["rays"])
    r = R.Renderer(sh_degree=int(g["sh_degree"]), white_background=True, fused=False)
    r.set_bg_color(torch.from_numpy(g["bg"]))
    leaves = {k: torch.from_numpy(g[f"in_{k}"]).clone().requires_grad_(True)
              for k in ("centers", "shs", "opacity", "scales", "rotations")}
    assert leaves["scales"].shape == (n, 2)
    ssp = torch.zeros(n, 4, requires_grad=True)
    rec = []
    st._Fn.record = rec
    out = r.render_img(cam, rays, leaves["centers"], leaves["shs"], leaves["opacity"], leaves["scales"],
                       leaves["rotations"], "cpu", depth_ratio=float(g["depth_ratio"]), screenspace_points=ssp)
    img_only = r.render_img(cam, None, *[v.detach() for v in leaves.values()], "cpu")
    st._Fn.record = None
    assert set(out) == set(OUT_KEYS)
    assert out["image"].shape == (h, w, 3) and out["depth"].shape == (h, w, 1) and out["acc_map"].shape == (h, w)
    assert out["rend_normal"].shape == (h, w, 3) and out["depth_normal"].shape == (h, w, 3) and out["rend_dist"].shape == (h, w)
    for k in OUT_KEYS:
        np.testing.assert_allclose(out[k].detach().numpy(), g[f"out_{k}"], rtol=0, atol=1e-6, err_msg=k)
    np.testing.assert_array_equal(img_only.detach().numpy(), g["image_only"])
    assert img_only.shape == (3, h, w)
    loss = surfel_loss(out, torch.from_numpy(g["target"]))
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=1e-6)
    grads = torch.autograd.grad(loss, list(leaves.values()) + [ssp])
    for k, gr in zip(list(leaves) + ["screenspace_points"], grads):
        ref = g[f"grad_{k}"]
        np.testing.assert_allclose(gr.numpy(), ref, rtol=1e-5, atol=1e-6 * np.abs(ref).max(), err_msg=k)
    assert g["grad_screenspace_points"].shape == (n, 4) and (g["grad_screenspace_points"][:, 2:] >= 0).all()
    o = rec[0]
    for k in ("radii", "point_list", "ranges", "tiles_touched", "rect"):
        np.testing.assert_array_equal(o[k], g[k], err_msg=k)
    np.testing.assert_array_equal(o["n_contrib"][0], g["n_contrib"])
    assert o["num_rendered"] == int(g["num_rendered"])


def test_depth_to_normal_of_a_tilted_plane():
    """Known answer: the depth map of a plane n.x = d seen through build_rays gives exactly that plane's normal."""
    from generativedensification_amd.camera import build_rays, look_at_c2w
    from generativedensification_amd.renderer_2dgs import depth_to_normal

    H, W = 40, 56
    c2w = look_at_c2w(torch.tensor([0.3, -0.2, 1.9]))
    rays = build_rays(c2w, 0.75, 0.75, H, W).double()
    n = torch.tensor([0.2, -0.3, 0.93], dtype=torch.float64)
    n = n / n.norm()
    o, d = rays[..., :3], rays[..., 3:]
    depth = ((0.1 - (o * n).sum(-1)) / (d * n).sum(-1))[None]   # ray-plane intersection parameter = view depth
    normal, pts = depth_to_normal(rays, depth)
    assert float(((pts * n).sum(-1) - 0.1).abs().max()) < 1e-12
    inner = normal[1:-1, 1:-1]
    assert float((inner - (-n if float((inner[0, 0] * n).sum()) < 0 else n)).abs().max()) < 1e-9
    assert float(normal[0].abs().max()) == 0 and float(normal[:, -1].abs().max()) == 0
