"""GPU diagnostic of the 2DGS surfel path: HIP vs the f32 / f64 oracle, field by field (prints, no asserts)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import util as U
from oracle.gdr_oracle import build

build()
for (N, H, W, seed, deg, sig) in [(3000, 128, 144, 1, 3, (0.0052, 0.02)), (20000, 250, 190, 2, 2, (0.0052, 0.00065)),
                                  (2000, 64, 64, 3, 0, (0.2,))]:
    case = U.make_surfel_case(N, H, W, seed, deg=deg, sigma0=sig, bg=(1.0, 0.5, 0.2))
    grads = U.rand_surfel_grads(case)
    hip, hg = U.run_surfel_hip(case, grads)
    o32, _ = U.run_surfel_oracle(case, "f32")
    o64, g64 = U.run_surfel_oracle(case, "f64", grads, nthreads=8)
    print(f"--- N={N} {H}x{W} deg={deg} sig={sig} D={hip['num_rendered']} (oracle {o32['num_rendered']})")
    for k in ("radii", "rect", "tiles_touched", "depths", "transMats", "xy", "normal_opacity", "rgb", "point_list", "ranges"):
        a, b = np.asarray(hip[k]), np.asarray(o32[k])
        if a.shape != b.shape:
            b = b.reshape(a.shape) if a.size == b.size else b
        same = a.shape == b.shape and np.array_equal(a.astype(b.dtype) if a.dtype != b.dtype else a, b)
        print(f"  {k:15s} bit-exact={same}" + ("" if same or a.shape != b.shape else f" maxdiff={np.abs(a.astype(np.float64)-b.astype(np.float64)).max():.3e} nbad={int((a!=b).sum())}"))
    nc = hip["n_contrib"].astype(np.int64)
    print("  n_contrib mismatch frac", float((nc[0] != o32["n_contrib"][0]).mean()), "median", float((nc[1] != o32["n_contrib"][1]).mean()))
    for ref, name in ((o32, "f32"), (o64, "f64")):
        print(f"  vs {name}: color rel_inf={U.rel_inf(hip['color'], ref['color']):.3e} outl={U.outlier_fraction(hip['color'], ref['color'], 1e-4, 1e-4):.2e}  PSNR={U.psnr(hip['color'], ref['color']):.1f}")
        for ch in range(7):
            print(f"      allmap[{ch}] rel_inf={U.rel_inf(hip['allmap'][ch], ref['allmap'][ch]):.3e} outl={U.outlier_fraction(hip['allmap'][ch], ref['allmap'][ch], 1e-4, 1e-4 * max(1e-30, np.abs(ref['allmap'][ch]).max())):.2e}")
    for k in ("means3D", "means2D", "shs", "opacities", "scales", "rotations"):
        if g64[k] is None:
            continue
        a, b = hg[k].reshape(g64[k].shape), g64[k]
        print(f"  grad {k:10s} rel_inf={U.rel_inf(a, b):.3e}  max|ref|={np.abs(b).max():.3e} outl(1e-3)={U.outlier_fraction(a, b, 1e-3, 1e-4 * np.abs(b).max()):.2e}")
