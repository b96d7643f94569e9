"""CPU: committed golden fixtures (tests/golden/*.npz, produced by make_golden.py with the
REFERENCE's MiniCam / Renderer / legacy render() on top of the oracle) vs
 (a) the repo's own host-side mirrors (camera.MiniCam, renderer.Renderer) and
 (b) the oracle as it builds today (regression)."""
import glob
import math
import os
import sys

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return dict(np.load(os.path.join(GOLD, name)))


@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLD, "minicam_*.npz"))))
def test_minicam_mirror_matches_reference_class(name):
    from generativedensification_amd.camera import MiniCam

    g = _load(name)
    cam = MiniCam(torch.from_numpy(g["c2w"]), int(g["width"]), int(g["height"]), torch.tensor(float(g["fovy"])),
                  torch.tensor(float(g["fovx"])), float(g["znear"]), float(g["zfar"]), "cpu")
    for k in ("world_view_transform", "full_proj_transform", "camera_center", "projection_matrix"):
        np.testing.assert_array_equal(getattr(cam, k).numpy(), g[k], err_msg=k)


def test_minicam_known_answer_from_survey():
    g = _load("minicam_identity.npz")
    np.testing.assert_allclose(g["world_view_transform"], [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 2, 1]])
    np.testing.assert_allclose(g["full_proj_transform"], [[2.5405, 0, 0, 0], [0, 2.5405, 0, 0], [0, 0, 1.25, 1], [0, 0, 1.875, 2]], atol=1e-4)
    np.testing.assert_array_equal(g["camera_center"], [0, 0, 2])  # sign quirk of lightning/utils.py:48


def _mirror_with_oracle(monkeypatch):
    """The repo's Renderer mirror with the oracle stand-in in place of the HIP rasterizer:
    isolates the HOST logic (activations, carrier, clamp, permutes) for a CPU check."""
    from generativedensification_amd import renderer as R
    from oracle.gdr_oracle import make_standin_module

    st = make_standin_module("f32")
    monkeypatch.setattr(R, "GaussianRasterizationSettings", st.GaussianRasterizationSettings)
    monkeypatch.setattr(R, "GaussianRasterizer", st.GaussianRasterizer)
    return R, st


@pytest.mark.parametrize("name", ["render_img_deg3.npz", "render_img_deg1.npz"])
def test_renderer_mirror_reproduces_reference_render_img(oracle_built, monkeypatch, name):
    from generativedensification_amd.camera import MiniCam
    from generativedensification_amd.synthetic import view_loss

    g = _load(name)
    R, st = _mirror_with_oracle(monkeypatch)
    cam = MiniCam(torch.from_numpy(g["c2w"]), int(g["w"]), int(g["h"]), torch.tensor(float(g["fov"])),
                  torch.tensor(float(g["fov"])), float(g["znear"]), float(g["zfar"]), "cpu")
    r = R.Renderer(sh_degree=int(g["sh_degree"]), white_background=True)
    r.set_bg_color(torch.from_numpy(g["bg"]))
    leaves = {k: torch.from_numpy(g[f"in_{k}"]).clone().requires_grad_(True)
              for k in ("centers", "shs", "opacity", "scales", "rotations")}
    ssp = torch.zeros(int(g["n"]), 4, requires_grad=True)
    rec = []
    st._Fn.record = rec
    out = r.render_img(cam, None, leaves["centers"], leaves["shs"], leaves["opacity"], leaves["scales"],
                       leaves["rotations"], "cpu", screenspace_points=ssp)
    st._Fn.record = None
    assert out["image"].shape == (int(g["h"]), int(g["w"]), 3) and out["depth"].shape == (int(g["h"]), int(g["w"]), 1)
    assert out["acc_map"].shape == (int(g["h"]), int(g["w"]))
    for k in ("image", "depth", "acc_map"):
        np.testing.assert_array_equal(out[k].detach().numpy(), g[k], err_msg=k)
    loss = view_loss(out, torch.from_numpy(g["target"]))
    grads = torch.autograd.grad(loss, list(leaves.values()) + [ssp])
    for k, gr in zip(list(leaves) + ["screenspace_points"], grads):
        np.testing.assert_allclose(gr.numpy(), g[f"grad_{k}"], rtol=1e-6, atol=1e-9, err_msg=k)
    assert g["grad_screenspace_points"].shape == (int(g["n"]), 4)
    assert (g["grad_screenspace_points"][:, 2:] >= 0).all() and g["grad_screenspace_points"][:, 2:].max() > 0
    o = rec[0]
    for k in ("radii", "point_list", "keys_sorted", "ranges", "n_contrib", "tiles_touched", "rect"):
        np.testing.assert_array_equal(o[k], g[k], err_msg=k)
    assert o["num_rendered"] == int(g["num_rendered"])


def test_legacy_caller_contract(oracle_built):
    """(N,3) means2D + colors_precomp + radii>0 visibility filter
    (lightning/point_decoder/layers/gaussian_renderer.py:88-114)."""
    from oracle.gdr_oracle import make_standin_module

    g = _load("legacy_render_colors.npz")
    st = make_standin_module("f32")
    n, h, w = int(g["n"]), int(g["h"]), int(g["w"])
    t = lambda k: torch.from_numpy(g[k]).clone().requires_grad_(True)
    pos, col, opa, sca, rot = t("position"), t("override_color"), t("opacity"), t("scaling"), t("rotation")
    ssp = torch.zeros(n, 3, requires_grad=True)
    rs = st.GaussianRasterizationSettings(
        image_height=h, image_width=w, tanfovx=math.tan(0.375), tanfovy=math.tan(0.375), bg=torch.from_numpy(g["bg"]),
        scale_modifier=1.0, viewmatrix=torch.from_numpy(g["world_view_transform"]),
        projmatrix=torch.from_numpy(g["full_proj_transform"]), sh_degree=0,
        campos=torch.from_numpy(g["camera_center"]), prefiltered=False, debug=False)
    img, radii, depth, alpha = st.GaussianRasterizer(rs)(means3D=pos, means2D=ssp, shs=None, colors_precomp=col,
                                                         opacities=opa, scales=sca, rotations=rot, cov3D_precomp=None)
    np.testing.assert_array_equal(img.detach().numpy(), g["render"])
    np.testing.assert_array_equal(radii.numpy(), g["radii"])
    np.testing.assert_array_equal((radii > 0).numpy(), g["visibility_filter"])
    grads = torch.autograd.grad((img * torch.from_numpy(g["grad_image"])).sum(), [pos, col, opa, sca, rot, ssp])
    for k, gr in zip(["position", "override_color", "opacity", "scaling", "rotation", "screenspace_points"], grads):
        np.testing.assert_allclose(gr.numpy(), g[f"grad_{k}"], rtol=1e-6, atol=1e-9, err_msg=k)
    assert grads[-1].shape == (n, 3) and not grads[-1][:, 2].any()
