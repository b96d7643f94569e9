#!/usr/bin/env python
"""tests/golden/make_golden.py — generates the committed golden fixtures (*.npz).

Runs ONLY in the authoring container (it imports the Python reference from
/root/reference, which never travels).  What is real reference code here:
  * `lightning/utils.py`  MiniCam / getProjectionMatrix          (matrix conventions)
  * `lightning/renderer.py`  Renderer.set_rasterizer / render_img (activations, (N,4)
    carrier, clamp, HWC permutes, settings construction)
  * `lightning/point_decoder/layers/gaussian_renderer.py` render() (legacy caller:
    colors_precomp / (N,3) means2D / bg on device)
The rasterizer underneath is NOT available in the reference (un-vendored submodule), so
the stand-in injected as `diff_gaussian_rasterization` is the oracle (oracle/gdr_oracle.py,
f32 build).  The fixtures therefore pin (a) the reference's caller-side conventions
exactly and (b) the oracle's numbers at generation time (regression vectors) — they do
not pin parity with the CUDA fork ("parity unpinned").

Fixtures are data only: inputs and expected outputs.  Usage: python tests/golden/make_golden.py
"""
import importlib.util
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle.gdr_oracle import make_standin_module  # noqa: E402

standin = make_standin_module("f32")
sys.modules["diff_gaussian_rasterization"] = standin
sys.path.insert(0, REF)
import lightning.renderer as ref_renderer  # noqa: E402  (reference code)
import lightning.utils as ref_utils  # noqa: E402  (reference code)

from generativedensification_amd.camera import look_at_c2w  # noqa: E402
from generativedensification_amd.synthetic import make_scene, make_targets, view_loss  # noqa: E402


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def cam_arrays(cam):
    return dict(world_view_transform=cam.world_view_transform, full_proj_transform=cam.full_proj_transform,
                camera_center=cam.camera_center, projection_matrix=cam.projection_matrix)


# ---- 1. MiniCam conventions (lightning/utils.py:22-48) --------------------------------------
def gen_minicam():
    # known answer quoted in SURVEY §8c
    c2w = torch.eye(4)
    c2w[2, 3] = -2.0
    cam = ref_utils.MiniCam(c2w, 64, 64, torch.tensor(0.75), torch.tensor(0.75), 0.5, 2.5, "cpu")
    save("minicam_identity.npz", c2w=c2w, width=64, height=64, fovy=0.75, fovx=0.75, znear=0.5, zfar=2.5,
         **cam_arrays(cam))
    g = torch.Generator().manual_seed(42)
    for k in range(3):
        eye = torch.randn(3, generator=g)
        eye = 1.9 * eye / eye.norm()
        c2w = look_at_c2w(eye)
        fovx, fovy = 0.6 + 0.1 * k, 0.75 - 0.05 * k
        w, h = [96, 128, 250][k], [64, 128, 190][k]
        cam = ref_utils.MiniCam(c2w, w, h, torch.tensor(fovy), torch.tensor(fovx), 1.1, 2.7, "cpu")
        save(f"minicam_lookat{k}.npz", c2w=c2w, width=w, height=h, fovy=fovy, fovx=fovx, znear=1.1, zfar=2.7,
             **cam_arrays(cam))


# ---- 2. Renderer.render_img (lightning/renderer.py:209-272) through the reference class ----
def gen_render_img(name, n, h, w, deg, sigma0, seed, bg, cam_eye):
    scene = make_scene(n, seed, sh_degree=deg, sigma0=sigma0)
    c2w = look_at_c2w(torch.tensor(cam_eye))
    cam = ref_utils.MiniCam(c2w, w, h, torch.tensor(0.75), torch.tensor(0.75), 1.1, 2.7, "cpu")
    r = ref_renderer.Renderer(sh_degree=deg, white_background=True)
    r.set_bg_color(torch.tensor(bg, dtype=torch.float32))
    leaves = {k: v.clone().requires_grad_(True) for k, v in scene.items()}
    ssp = torch.zeros(n, 4, requires_grad=True)
    rec = []
    standin._Fn.record = rec
    out = r.render_img(cam, None, leaves["centers"], leaves["shs"], leaves["opacity"], leaves["scales"],
                       leaves["rotations"], "cpu", screenspace_points=ssp)
    standin._Fn.record = None
    target = make_targets(1, h, w, seed)[0]
    loss = view_loss(out, target)
    grads = torch.autograd.grad(loss, list(leaves.values()) + [ssp])
    o = rec[0]
    save(name, n=n, h=h, w=w, sh_degree=deg, bg=np.asarray(bg, np.float32), c2w=c2w, fov=0.75, znear=1.1, zfar=2.7,
         **{f"in_{k}": v for k, v in scene.items()}, **cam_arrays(cam), target=target,
         image=out["image"], depth=out["depth"], acc_map=out["acc_map"], loss=loss,
         **{f"grad_{k}": g for k, g in zip(list(leaves) + ["screenspace_points"], grads)},
         radii=o["radii"], num_rendered=o["num_rendered"], point_list=o["point_list"],
         keys_sorted=o["keys_sorted"], ranges=o["ranges"], n_contrib=o["n_contrib"],
         tiles_touched=o["tiles_touched"], rect=o["rect"])


# ---- 3. legacy caller (point_decoder/layers/gaussian_renderer.py:17-114) --------------------
def gen_legacy(name, n, h, w, seed):
    spec = importlib.util.spec_from_file_location(
        "ref_gaussian_renderer", os.path.join(REF, "lightning/point_decoder/layers/gaussian_renderer.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    scene = make_scene(n, seed, sh_degree=0, sigma0=(0.02,))
    c2w = look_at_c2w(torch.tensor([1.2, -1.0, 1.0]))
    cam = ref_utils.MiniCam(c2w, w, h, torch.tensor(0.75), torch.tensor(0.75), 1.1, 2.7, "cpu")
    pos = scene["centers"].clone().requires_grad_(True)
    col = torch.sigmoid(scene["shs"][:, 0, :]).clone().requires_grad_(True)
    opa = torch.sigmoid(scene["opacity"]).clone().requires_grad_(True)
    sca = torch.exp(scene["scales"]).clone().requires_grad_(True)
    rot = torch.nn.functional.normalize(scene["rotations"]).clone().requires_grad_(True)
    ssp = torch.zeros(n, 3, requires_grad=True)
    bg = torch.tensor([0.2, 0.4, 0.6])
    pkg = mod.render(0.75, 0.75, w, h, cam.world_view_transform, cam.full_proj_transform, cam.camera_center,
                     pos, None, opa, sca, rot, ssp, bg, 0, override_color=col)
    g = torch.Generator().manual_seed(seed)
    gimg = torch.randn(3, h, w, generator=g)
    loss = (pkg["render"] * gimg).sum()
    grads = torch.autograd.grad(loss, [pos, col, opa, sca, rot, ssp])
    save(name, n=n, h=h, w=w, bg=bg, c2w=c2w, fov=0.75, **cam_arrays(cam), position=pos, override_color=col,
         opacity=opa, scaling=sca, rotation=rot, grad_image=gimg, render=pkg["render"],
         visibility_filter=pkg["visibility_filter"], radii=pkg["radii"],
         **{f"grad_{k}": v for k, v in zip(["position", "override_color", "opacity", "scaling", "rotation",
                                             "screenspace_points"], grads)})


if __name__ == "__main__":
    gen_minicam()
    gen_render_img("render_img_deg3.npz", 1500, 80, 112, 3, (0.03, 0.008), 101, (1.0, 1.0, 1.0), [1.5, 0.9, 0.7])
    gen_render_img("render_img_deg1.npz", 2500, 64, 64, 1, (0.0052, 0.02), 102, (0.5, 0.5, 0.5), [-1.2, 1.3, -0.6])
    gen_legacy("legacy_render_colors.npz", 1200, 48, 72, 103)
