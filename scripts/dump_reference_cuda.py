#!/usr/bin/env python
"""ONE-SHOT PINNING SCRIPT — to be run by a maintainer ON A CUDA BOX where the reference's real extension is installed
(`pip3 install ./third_party/diff-gaussian-rasterization`, /root/reference/README.md:49-53); it cannot run here (the
submodule is empty in /root/reference and there is no CUDA).  It feeds the fork the SAME seeded inputs our fixtures use
and dumps everything parity depends on into one .npz that drops into tests/golden/:

    python scripts/dump_reference_cuda.py --out tests/golden/cuda_reference_c1.npz

Recorded: colour / radii / depth / alpha of `GaussianRasterizer(...)`, and the gradients of a fixed scalar (random
upstream gradients, seeded) w.r.t. means3D, the (N,4) means2D carrier, shs, opacities, scales, rotations.  From these
the open conventions listed in INTEGRATION.md §0 are decided by data, not by lineage:
  R1  does dL/d depth reach means3D?        -> compare means3D grads with / without a depth upstream gradient
  R3  what do columns 2:4 of the carrier hold -> compared with the oracle's sum |per-pixel term|
  R4  is depth sum_i w_i z_i or normalised    -> depth / alpha against the oracle's two variants
  tie order, 0.99 clamp, 1/255 and 1e-4 thresholds -> n_contrib-equivalent: image exactness on threshold pixels
`tests/test_cuda_reference.py` picks every tests/golden/cuda_reference_*.npz up when present: the oracle (CPU suite) and the
HIP path (-m gpu) are compared with it under north_star's bars, and the test prints which of R1 / R3 / R4 the data selects
(tests/cuda_reference.py).  With no file both skip, naming this script.

`--standin oracle` runs the SAME dump code on the repo's oracle stand-in (CPU, no CUDA): that is how the consumer is tested
here (a synthetic dump in a temporary directory — never committed as a cuda_reference_* fixture, it pins nothing).
"""
import argparse
import math

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--n", type=int, default=10_000)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--deg", type=int, default=3)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--standin", choices=["oracle"], default=None,
                    help="self-test of the consumer: dump the repo's oracle stand-in (CPU) instead of the CUDA extension")
    a = ap.parse_args()
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if a.standin:
        from oracle.gdr_oracle import make_standin_module
        mod = make_standin_module("f32", nthreads=os.cpu_count() or 1)
        GaussianRasterizationSettings, GaussianRasterizer = mod.GaussianRasterizationSettings, mod.GaussianRasterizer
        dev, source = torch.device("cpu"), "oracle-standin (synthetic: pins nothing)"
    else:
        from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # the REAL extension
        import diff_gaussian_rasterization as _ext
        if "generativedensification_amd" in (getattr(_ext, "__file__", "") or "") or hasattr(_ext, "__oracle_standin__"):
            raise SystemExit("this is the MI355X drop-in package, not the reference's CUDA extension: nothing to pin")
        dev, source = torch.device("cuda"), "cuda"
    from generativedensification_amd.camera import orbit_cameras      # pure torch: same cameras / scene as our fixtures
    from generativedensification_amd.synthetic import make_scene

    sc = make_scene(a.n, a.seed, sh_degree=a.deg)
    cam = orbit_cameras(4, a.size, a.size)[1]
    leaves = dict(means3D=sc["centers"], shs=sc["shs"], opacities=torch.sigmoid(sc["opacity"]),
                  scales=torch.exp(sc["scales"]), rotations=torch.nn.functional.normalize(sc["rotations"]))
    leaves = {k: v.to(dev).requires_grad_(True) for k, v in leaves.items()}
    m2d = torch.zeros(a.n, 4, device=dev, requires_grad=True)
    rs = GaussianRasterizationSettings(
        image_height=a.size, image_width=a.size, tanfovx=math.tan(0.375), tanfovy=math.tan(0.375),
        bg=torch.ones(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dev),
        projmatrix=cam.full_proj_transform.to(dev), sh_degree=a.deg, campos=cam.camera_center.to(dev),
        prefiltered=False, debug=False)
    color, radii, depth, alpha = GaussianRasterizer(rs)(means2D=m2d, **leaves)
    g = torch.Generator().manual_seed(123)
    gc, gd, ga = (torch.randn(c, a.size, a.size, generator=g).to(dev) for c in (3, 1, 1))
    out = {"color": color, "radii": radii, "depth": depth, "alpha": alpha}
    for tag, scalar in (("all", (color * gc).sum() + (depth * gd).sum() + (alpha * ga).sum()),
                        ("color_only", (color * gc).sum()), ("depth_only", (depth * gd).sum())):
        grads = torch.autograd.grad(scalar, list(leaves.values()) + [m2d], retain_graph=True, allow_unused=True)
        for k, v in zip(list(leaves) + ["means2D"], grads):
            out[f"grad_{tag}_{k}"] = torch.zeros(1) if v is None else v
    np.savez_compressed(a.out, **{k: v.detach().cpu().numpy() for k, v in out.items()},
                        upstream_color=gc.cpu().numpy(), upstream_depth=gd.cpu().numpy(), upstream_alpha=ga.cpu().numpy(),
                        n=a.n, size=a.size, deg=a.deg, seed=a.seed, source=source)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
