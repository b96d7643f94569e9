#!/usr/bin/env python
"""The literal "PyTorch-CPU" baseline named by BASELINE.json's north_star, timed for bench.py in a child process (so that a
pathological host cannot stall the bench: bench.py gives it a wall-clock limit).  oracle/torch_ref.py = vectorised torch
forward + autograd backward; one view of the workload's image size on the first n_s Gaussians of the seeded scene.
Prints one JSON object.  TEST / MEASUREMENT INFRASTRUCTURE (imports oracle/)."""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, required=True)
ap.add_argument("--n-sample", type=int, default=20_000)
ap.add_argument("--seed", type=int, required=True)
ap.add_argument("--sigma0", type=str, required=True)
ap.add_argument("--h", type=int, required=True)
ap.add_argument("--w", type=int, required=True)
ap.add_argument("--deg", type=int, required=True)
ap.add_argument("--layout", default="cube")
ap.add_argument("--threads", type=int, default=16)
a = ap.parse_args()

from generativedensification_amd.camera import orbit_cameras  # noqa: E402
from generativedensification_amd.synthetic import make_scene, make_targets  # noqa: E402
from oracle import torch_ref as TR  # noqa: E402

torch.set_num_threads(a.threads)
sig = tuple(float(x) for x in a.sigma0.split(","))
sc = make_scene(a.n, a.seed, sh_degree=a.deg, sigma0=sig, layout=a.layout)
n_s = min(a.n, a.n_sample)
c = {k: v[:n_s].contiguous() for k, v in sc.items()}
cam = orbit_cameras(4, a.w, a.h)[0]
tg = make_targets(1, a.h, a.w, a.seed)[0].permute(2, 0, 1)
kw = dict(image_height=a.h, image_width=a.w, tanfovx=math.tan(0.375), tanfovy=math.tan(0.375), bg=torch.ones(3),
          scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=a.deg,
          campos=cam.camera_center)
t0 = time.perf_counter()
lv = {k: v.clone().requires_grad_(True) for k, v in c.items()}
color, _, depth, alpha = TR.render(lv["centers"], torch.sigmoid(lv["opacity"]).reshape(-1), shs=lv["shs"],
                                   scales=torch.exp(lv["scales"]), rotations=torch.nn.functional.normalize(lv["rotations"]), **kw)
(((color.clamp(0, 1) - tg) ** 2).mean() + 0.1 * depth.mean() + 0.1 * alpha.mean()).backward()
tt = time.perf_counter() - t0
print(json.dumps(dict(seconds=tt, n_sample=n_s, threads=torch.get_num_threads())))
