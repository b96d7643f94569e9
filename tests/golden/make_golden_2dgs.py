#!/usr/bin/env python
"""tests/golden/make_golden_2dgs.py — golden fixtures of the 2DGS surfel path (render2dgs_*.npz).

Runs ONLY in the authoring container: it imports the reference's 2DGS adaptor
(/root/reference/lightning/renderer_2dgs.py — real reference code: activations, (N,4) carrier, allmap slicing,
normal rotation, expected/median depth mix, depth_to_normal) and MiniCam (lightning/utils.py).  The two packages
that file imports and the reference does not contain are injected as stand-ins: `diff_surfel_rasterization` = the
CPU oracle (oracle/gsr_oracle.py, f32 build), `simple_knn._C.distCUDA2` = a numpy stub (never called by render_img).
The fixtures pin the reference's caller-side conventions exactly and the oracle's numbers at generation time; they do
not pin parity with the original CUDA package ("parity unpinned").  Data only: inputs and expected outputs.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle.gsr_oracle import make_simple_knn_stub, make_surfel_standin_module  # noqa: E402

standin = make_surfel_standin_module("f32")
sys.modules["diff_surfel_rasterization"] = standin
sys.modules["simple_knn"], sys.modules["simple_knn._C"] = make_simple_knn_stub()
sys.path.insert(0, REF)
import lightning.renderer_2dgs as ref_2dgs  # noqa: E402  (reference code)
import lightning.utils as ref_utils  # noqa: E402  (reference code)

from generativedensification_amd.camera import build_rays, look_at_c2w  # noqa: E402
from generativedensification_amd.synthetic import make_scene, make_targets, surfel_loss  # noqa: E402


def save(name, **arrays):
    out = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()}
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def gen(name, n, h, w, deg, sigma0, seed, bg, cam_eye, depth_ratio):
    scene = make_scene(n, seed, sh_degree=deg, sigma0=sigma0)
    scene["scales"] = scene["scales"][:, :2].contiguous()
    c2w = look_at_c2w(torch.tensor(cam_eye))
    cam = ref_utils.MiniCam(c2w, w, h, torch.tensor(0.75), torch.tensor(0.75), 1.1, 2.7, "cpu")
    rays = build_rays(c2w, 0.75, 0.75, h, w)
    r = ref_2dgs.Renderer(sh_degree=deg, white_background=True)
    r.set_bg_color(torch.tensor(bg, dtype=torch.float32))
    leaves = {k: v.clone().requires_grad_(True) for k, v in scene.items()}
    ssp = torch.zeros(n, 4, requires_grad=True)
    rec = []
    standin._Fn.record = rec
    out = r.render_img(cam, rays, leaves["centers"], leaves["shs"], leaves["opacity"], leaves["scales"],
                       leaves["rotations"], "cpu", depth_ratio=depth_ratio, screenspace_points=ssp)
    img_only = r.render_img(cam, None, scene["centers"], scene["shs"], scene["opacity"], scene["scales"],
                            scene["rotations"], "cpu")
    standin._Fn.record = None
    target = make_targets(1, h, w, seed)[0]
    loss = surfel_loss(out, target)
    grads = torch.autograd.grad(loss, list(leaves.values()) + [ssp])
    o = rec[0]
    save(name, n=n, h=h, w=w, sh_degree=deg, bg=np.asarray(bg, np.float32), c2w=c2w, fov=0.75, znear=1.1, zfar=2.7,
         depth_ratio=depth_ratio, rays=rays, world_view_transform=cam.world_view_transform,
         full_proj_transform=cam.full_proj_transform, camera_center=cam.camera_center,
         **{f"in_{k}": v for k, v in scene.items()}, target=target, image_only=img_only,
         **{f"out_{k}": v for k, v in out.items()}, loss=loss,
         **{f"grad_{k}": g for k, g in zip(list(leaves) + ["screenspace_points"], grads)},
         radii=o["radii"], num_rendered=o["num_rendered"], point_list=o["point_list"], ranges=o["ranges"],
         n_contrib=o["n_contrib"][0], tiles_touched=o["tiles_touched"], rect=o["rect"])


if __name__ == "__main__":
    gen("render2dgs_deg3.npz", 1200, 72, 96, 3, (0.03, 0.01), 201, (1.0, 1.0, 1.0), [1.5, 0.9, 0.7], 0.0)
    gen("render2dgs_deg1_median.npz", 2000, 64, 64, 1, (0.0052, 0.03), 202, (0.5, 0.5, 0.5), [-1.2, 1.3, -0.6], 1.0)
