"""-m gpu: HIP product path (through the C ABI) vs the CPU oracle on the same seeded
inputs.  Integer / index work must be BIT-EXACT; floats within 1e-4 relative
(BASELINE.json north_star), tolerance written at each assert."""
import numpy as np
import pytest
import torch

import util as U

pytestmark = pytest.mark.gpu

CASES = [
    # N, H, W, seed, deg, sigma0
    (10_000, 256, 256, 0, 3, (0.0052, 0.00065)),     # C1 of BASELINE.json
    (3_000, 250, 190, 3, 3, (0.03, 0.01)),           # non-multiple-of-16 image, long tile lists
    (20_000, 128, 128, 5, 1, (0.02,)),               # SH degree 1 (reference default, configs/base.yaml:14)
    (2_000, 64, 64, 7, 0, (0.05,)),                  # SH degree 0, heavy overlap (early termination)
    (5_000, 96, 160, 9, 2, (0.01, 0.002)),           # SH degree 2, M = 9 (unaligned SH rows)
]


def _check_forward(o, h):
    # ---- integers / indices: bit-exact --------------------------------------------
    assert h["num_rendered"] == o["num_rendered"]
    np.testing.assert_array_equal(h["radii"], o["radii"])
    np.testing.assert_array_equal(h["rect"], o["rect"])
    np.testing.assert_array_equal(h["tiles_touched"].astype(np.uint32), o["tiles_touched"])
    np.testing.assert_array_equal(h["keys_sorted"].view(np.uint64), o["keys_sorted"])
    np.testing.assert_array_equal(h["point_list"].view(np.uint32), o["point_list"])
    np.testing.assert_array_equal(h["ranges"].view(np.uint32), o["ranges"])
    cl = np.stack([(h["clamped"] >> k) & 1 for k in range(3)], 1)
    np.testing.assert_array_equal(cl, o["clamped"])
    # ---- per-Gaussian floats: same op order, no FMA contraction -> bit-exact ---------
    for k in ("depths", "xy", "conic_opacity", "cov3D"):
        np.testing.assert_array_equal(h[k], o[k], err_msg=k)
    np.testing.assert_array_equal(h["rgb"][:, :3], o["rgb"])
    # ---- rendered images: 1e-4 relative (exp ulp differences can flip a 1/255 or 1e-4
    #      threshold at isolated pixels; those are counted, not hidden) ------------------
    for k in ("color", "depth", "alpha"):
        assert U.rel_inf(h[k], o[k]) < 5e-3, k                      # worst single pixel (flip-sized)
        assert U.outlier_fraction(h[k], o[k], rtol=1e-4, atol=1e-5) < 1e-4, k  # 1e-4 relative bar
    assert U.psnr(np.clip(h["color"], 0, 1), np.clip(o["color"], 0, 1)) > 60.0
    assert (h["n_contrib"].view(np.uint32) != o["n_contrib"]).mean() < 1e-4
    assert U.outlier_fraction(h["final_T"], o["final_T"], rtol=1e-4, atol=1e-6) < 1e-4


@pytest.mark.parametrize("N,H,W,seed,deg,sigma0", CASES)
def test_forward_and_backward_vs_oracle(oracle_built, N, H, W, seed, deg, sigma0):
    case = U.make_case(N, H, W, seed, deg=deg, sigma0=sigma0)
    grads = U.rand_grads(case)
    o, og = U.run_oracle(case, "f32", grads)
    h, hg = U.run_hip(case, grads)
    _check_forward(o, h)
    _, og64 = U.run_oracle(case, "f64", grads)
    for k in ("means3D", "means2D", "shs", "opacities", "scales", "rotations"):
        ref = og64[k].reshape(hg[k].shape)
        e_hip = U.rel_inf(hg[k], ref)
        e_f32 = U.rel_inf(og[k].reshape(hg[k].shape), ref)
        # 1e-4 relative (north_star); also no worse than 10x the f32 oracle's own rounding error
        assert e_hip < 1e-4 or e_hip < 10 * e_f32, (k, e_hip, e_f32)
    # culled Gaussians get exact zeros everywhere (set_detect_anomaly-safe, train_lightning.py:31)
    culled = o["radii"] == 0
    for k in ("means3D", "means2D", "shs", "opacities", "scales", "rotations"):
        assert np.isfinite(hg[k]).all(), k
        assert not hg[k][culled].any(), k


def test_colors_precomp_and_cov3d_precomp(oracle_built):
    case = U.make_case(4_000, 112, 144, 21, deg=0, sigma0=(0.02, 0.004), colors_precomp=True, cov_precomp=True)
    grads = U.rand_grads(case)
    o, og = U.run_oracle(case, "f32", grads)
    h, hg = U.run_hip(case, grads)
    assert h["num_rendered"] == o["num_rendered"]
    np.testing.assert_array_equal(h["point_list"].view(np.uint32), o["point_list"])
    assert U.outlier_fraction(h["color"], o["color"], 1e-4, 1e-5) < 1e-4
    _, og64 = U.run_oracle(case, "f64", grads)
    for k in ("means3D", "means2D", "colors_precomp", "opacities", "cov3D_precomp"):
        ref = og64[k].reshape(hg[k].shape)
        e_hip, e_f32 = U.rel_inf(hg[k], ref), U.rel_inf(og[k].reshape(hg[k].shape), ref)
        assert e_hip < 1e-4 or e_hip < 10 * e_f32, (k, e_hip, e_f32)
    assert hg["shs"] is None and hg["scales"] is None and hg["rotations"] is None


def test_empty_and_all_culled(oracle_built):
    dev = torch.device("cuda:0")
    from generativedensification_amd.rasterizer import GaussianRasterizer

    case = U.make_case(100, 64, 64, 1, deg=1, sigma0=(0.02,))
    rs = U.settings_torch(case, dev)
    r = GaussianRasterizer(rs)
    # all behind the camera
    means = (case["means3D"] * 0 + case["campos"] * 0 - 100.0 * torch.tensor([0.0, 0.0, 0.0])).to(dev)
    means = (case["means3D"] + 50.0 * (case["view"][:3, 2])).to(dev) * -1.0
    means.requires_grad_(True)
    col, radii, dep, alp = r(means3D=means, means2D=torch.zeros(100, 4, device=dev, requires_grad=True),
                             shs=case["shs"].to(dev), opacities=case["opacities"].to(dev),
                             scales=case["scales"].to(dev), rotations=case["rotations"].to(dev))
    (col.sum() + dep.sum() + alp.sum()).backward()
    if int((radii > 0).sum()) == 0:
        assert torch.allclose(col, rs.bg[:, None, None].expand_as(col))
        assert float(alp.detach().abs().max()) == 0.0
        assert float(means.grad.abs().max()) == 0.0
    # N = 0
    z = torch.zeros(0, 3, device=dev)
    col, radii, dep, alp = r(means3D=z, means2D=torch.zeros(0, 4, device=dev), shs=torch.zeros(0, 4, 3, device=dev),
                             opacities=torch.zeros(0, 1, device=dev), scales=z, rotations=torch.zeros(0, 4, device=dev))
    assert radii.numel() == 0 and torch.allclose(col, rs.bg[:, None, None].expand_as(col))
