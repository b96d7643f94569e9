"""CPU (-m "not gpu"): the 2DGS surfel oracle (oracle/gsr_oracle.c) checked against what can pin it here — an
independent autograd restatement, closed-form known answers and edge cases.  (`diff_surfel_rasterization` is absent
from /root/reference and the reference holds no tests for it: parity unpinned)"""
import math

import numpy as np
import pytest
import torch

import util as U
from oracle import torch_ref
from oracle import torch_ref_surfel as TS
from oracle.gdr_oracle import Settings
from oracle.gsr_oracle import SurfelOracle


def _surfel_case(N, H, W, seed, **kw):
    case = U.make_case(N, H, W, seed, **kw)
    case["scales"] = case["scales"][:, :2].contiguous()
    return case


def _front_settings(H, W, bg=(0.0, 0.0, 0.0), fov=0.75, deg=0, dist=2.0):
    from generativedensification_amd.camera import MiniCam

    c2w = torch.eye(4)
    c2w[2, 3] = -dist
    cam = MiniCam(c2w, W, H, torch.tensor(fov), torch.tensor(fov), 0.5, 10.0, "cpu")
    t = math.tan(fov / 2)
    return Settings(H, W, t, t, np.array(bg, np.float64), 1.0, cam.world_view_transform.numpy(),
                    cam.full_proj_transform.numpy(), deg, cam.camera_center.numpy())


@pytest.mark.parametrize("sigma0,precomp", [((0.03, 0.01), False), ((0.004,), False), ((0.03,), True)])
def test_c_surfel_oracle_matches_autograd(oracle_built, sigma0, precomp):
    """forward to 1e-12, every input gradient to 1e-10 relative, the raw dL/dT of the render stage and the
    densification signal means2D[:, :2] = dL/dT[.][2] * depth * 0.5 * (W|H).  sigma0 = 0.004 puts most surfels on the
    screen-space low-pass branch (rho2d < rho3d); precomp: transMat_precomp + colors_precomp inputs."""
    case = _surfel_case(220, 48, 40, 5, deg=3, sigma0=sigma0, bg=(1.0, 0.5, 0.2))
    dt = torch.float64
    o = SurfelOracle("f64")
    s = U.settings_np(case)
    kw = torch_ref.settings_kwargs(s)
    if precomp:
        T, _ = TS.transmats(case["means3D"].to(dt), case["scales"].to(dt), case["rotations"].to(dt), 1.0,
                            kw["projmatrix"], 40, 48)
        ins = dict(means3D=case["means3D"].to(dt), opacities=case["opacities"].to(dt), transMat_precomp=T.reshape(-1, 9),
                   colors_precomp=torch.sigmoid(case["shs"][:, 0, :]).to(dt))
    else:
        ins = {k: case[k].to(dt) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    ins = {k: v.clone().requires_grad_(True) for k, v in ins.items()}
    probe = {}
    targs = {k: v for k, v in ins.items() if k not in ("means3D", "opacities")}
    c, r, am = TS.render(ins["means3D"], ins["opacities"], probe=probe, **targs, **kw)
    out = o.forward(ins["means3D"].detach().numpy(), ins["opacities"].detach().numpy(), s,
                    **{k: v.detach().numpy() for k, v in targs.items()})
    assert out["num_rendered"] > 100
    assert np.abs(c.detach().numpy() - out["color"]).max() < 1e-12
    assert np.abs(am.detach().numpy() - out["allmap"]).max() < 1e-12
    np.testing.assert_array_equal(r.numpy(), out["radii"])
    g = torch.Generator().manual_seed(1)
    gc = torch.randn(3, 48, 40, generator=g, dtype=dt)
    ga = torch.randn(7, 48, 40, generator=g, dtype=dt)
    gt = torch.autograd.grad((c * gc).sum() + (am * ga).sum(), list(ins.values()), allow_unused=True)
    og = o.backward(out, gc.numpy(), ga.numpy())
    for k, gg in zip(ins, gt):
        ref = np.zeros(tuple(ins[k].shape)) if gg is None else gg.numpy()  # means3D is unused with both precomps
        got = og[k].reshape(ref.shape)
        assert np.abs(got - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max()), k
    Tg = probe["T_ray"].grad.reshape(-1, 9).numpy()
    assert np.abs(Tg - og["_partial"]["transMat"]).max() <= 1e-10 * max(1.0, np.abs(Tg).max())
    vis, dep, m2 = out["radii"] > 0, probe["depth"].numpy(), og["means2D"]
    assert np.abs(m2[vis, 0] - Tg[vis, 2] * dep[vis] * 0.5 * 40).max() <= 1e-9 * max(1.0, np.abs(m2).max())
    assert np.abs(m2[vis, 1] - Tg[vis, 5] * dep[vis] * 0.5 * 48).max() <= 1e-9 * max(1.0, np.abs(m2).max())
    assert (m2[:, 2:] >= 0).all() and (m2[:, 2:] + 1e-12 >= np.abs(m2[:, :2])).all()  # sum|x| >= |sum x|
    if sigma0 == (0.004,):
        assert np.abs(og["_partial"]["mean2D"][:, :2]).max() > 0  # the low-pass branch was exercised


def test_abs_channels_are_sums_of_per_pixel_abs_terms(oracle_built):
    case = _surfel_case(10, 16, 16, 3, deg=1, sigma0=(0.25,))
    dt = torch.float64
    o = SurfelOracle("f64")
    s = U.settings_np(case)
    ins = [case[k].to(dt) for k in ("means3D", "opacities", "shs", "scales", "rotations")]
    out = o.forward(*[a.numpy() for a in ins[:2]], s, shs=ins[2].numpy(), scales=ins[3].numpy(), rotations=ins[4].numpy())
    g = torch.Generator().manual_seed(2)
    gc, ga = torch.randn(3, 16, 16, generator=g, dtype=dt), torch.randn(7, 16, 16, generator=g, dtype=dt)
    og = o.backward(out, gc.numpy(), ga.numpy())
    probe = {}
    c, r, am = TS.render(ins[0], ins[1], shs=ins[2], scales=ins[3], rotations=ins[4], probe=probe,
                         **torch_ref.settings_kwargs(s))
    per_pix = (c * gc).sum(0) + (am * ga).sum(0)
    acc = torch.zeros(10, 2, dtype=dt)
    for y in range(16):
        for x in range(16):
            (gT,) = torch.autograd.grad(per_pix[y, x], probe["T_ray"], retain_graph=True)
            acc += torch.stack([gT[:, 0, 2], gT[:, 1, 2]], 1).abs()
    ref = acc.numpy() * probe["depth"].numpy()[:, None] * 0.5 * 16
    assert float(ref.max()) > 0
    assert np.abs(og["means2D"][:, 2:4] - ref).max() <= 1e-9 * max(1.0, float(ref.max()))


def test_known_answer_single_fronto_parallel_surfel(oracle_built):
    """One surfel at the origin facing the camera (identity rotation, scale s): along the image row through the
    centre u = x_world / s, so alpha(pixel) = o * exp(-u^2/2) wherever the object-space term beats the low-pass floor;
    allmap depth = alpha * 2, normal = alpha * (0,0,-1) (flipped towards the camera), median depth = 2."""
    H = W = 64
    s = _front_settings(H, W, bg=(0.2, 0.3, 0.4))
    o = SurfelOracle("f64")
    sc, op = 0.2, 0.8
    out = o.forward(np.zeros((1, 3)), np.array([op]), s, colors_precomp=np.array([[1.0, 0.5, 0.25]]),
                    scales=np.array([[sc, sc]]), rotations=np.array([[1.0, 0, 0, 0]]))
    assert out["radii"][0] > 0
    f = W / (2 * s.tanfovx)  # pixel = f * x / z + (W-1)/2 at z = 2
    y = 32
    for x in (31, 32, 36, 40, 45):
        xw = (x - 31.5) * 2.0 / f
        yw = (y - 31.5) * 2.0 / f
        a = op * math.exp(-0.5 * (xw * xw + yw * yw) / (sc * sc))
        assert abs(out["allmap"][1, y, x] - a) < 2e-6
        assert abs(out["allmap"][0, y, x] - 2.0 * a) < 2e-6
        np.testing.assert_allclose(out["allmap"][2:5, y, x], [0, 0, -a], atol=2e-6)
        assert abs(out["allmap"][5, y, x] - 2.0) < 2e-6
        assert abs(out["allmap"][6, y, x]) < 1e-12  # one contributor: no distortion
        np.testing.assert_allclose(out["color"][:, y, x], a * np.array([1.0, 0.5, 0.25]) + (1 - a) * np.array(s.bg), atol=2e-6)


def test_low_pass_floor_and_two_surfel_distortion(oracle_built):
    """(i) A surfel far smaller than a pixel still renders through the screen-space low-pass: alpha at the nearest
    pixel = o * exp(-|d|^2) (FilterInvSquare = 2), depth = centre depth.  (ii) Two stacked opaque-ish surfels:
    distortion = w0 w1 (m0 - m1)^2 with m = far/(far-near) (1 - near/z)."""
    H = W = 32
    s = _front_settings(H, W)
    o = SurfelOracle("f64")
    out = o.forward(np.zeros((1, 3)), np.array([0.9]), s, colors_precomp=np.ones((1, 3)),
                    scales=np.array([[1e-5, 1e-5]]), rotations=np.array([[1.0, 0, 0, 0]]))
    a = 0.9 * math.exp(-(0.5 ** 2 + 0.5 ** 2))  # centre at (15.5, 15.5): nearest pixels are 0.5 px away on each axis
    assert abs(out["allmap"][1, 15, 15] - a) < 2e-6 and abs(out["allmap"][1, 16, 16] - a) < 2e-6
    assert abs(out["allmap"][0, 15, 15] - 2.0 * a) < 2e-6
    z0, z1 = 2.0, 2.5
    out = o.forward(np.array([[0, 0, 0.0], [0, 0, z1 - z0]]), np.array([0.6, 0.7]), s, colors_precomp=np.ones((2, 3)),
                    scales=np.full((2, 2), 0.5), rotations=np.array([[1.0, 0, 0, 0]] * 2))
    m = lambda z: 100.0 / (100.0 - 0.2) * (1 - 0.2 / z)
    y = x = 16
    a0, a1 = out["allmap"][1, y, x], None
    # alphas at this pixel from the closed form of the previous test
    f = W / (2 * s.tanfovx)
    r2 = lambda z: 2 * ((0.5 * z / f) ** 2) / 0.25
    al0, al1 = 0.6 * math.exp(-0.5 * r2(z0)), 0.7 * math.exp(-0.5 * r2(z1))
    w0, w1 = al0, al1 * (1 - al0)
    assert abs(out["allmap"][1, y, x] - (w0 + w1)) < 2e-6
    assert abs(out["allmap"][6, y, x] - w0 * w1 * (m(z0) - m(z1)) ** 2) < 1e-9
    assert abs(out["allmap"][5, y, x] - (z1 if (1 - al0) > 0.5 else z0)) < 2e-6


def test_backfacing_flip_near_plane_and_rect_edges(oracle_built):
    s = _front_settings(50, 38, dist=2.0)  # non multiple of 16
    o = SurfelOracle("f32")
    q_front, q_back = [1.0, 0, 0, 0], [0.0, 1.0, 0, 0]  # 180 deg about x: normal (0,0,-1) vs (0,0,+1)
    means = np.array([[0, 0, 0.0], [0, 0, 0.0], [0, 0, -1.81], [0, 0, -1.79], [5.0, 0, 0]])
    out = o.forward(means, np.full(5, 0.5), s, colors_precomp=np.ones((5, 3)), scales=np.full((5, 2), 0.05),
                    rotations=np.array([q_front, q_back, q_front, q_front, q_front]))
    np.testing.assert_allclose(out["normal_opacity"][0, :3], out["normal_opacity"][1, :3], atol=1e-6)  # both face the camera
    assert out["normal_opacity"][0, 2] < 0
    assert out["radii"][2] == 0 and out["radii"][3] > 0   # view z = 0.19 culled, 0.21 kept
    assert out["radii"][4] == 0 and out["tiles_touched"][4] == 0  # off screen: empty tile rect
    gx, gy = 3, 4
    assert (out["rect"][:, 2] <= gx).all() and (out["rect"][:, 3] <= gy).all()
    assert out["tiles_touched"][3] == gx * gy  # the close-up surfel covers the whole 38x50 image


def test_openmp_threads_do_not_change_surfel_results_beyond_rounding(oracle_built):
    case = _surfel_case(400, 64, 48, 9, deg=2, sigma0=(0.02,))
    s = U.settings_np(case)
    a = [case[k].numpy() for k in ("means3D", "opacities")]
    kw = dict(shs=case["shs"].numpy(), scales=case["scales"].numpy(), rotations=case["rotations"].numpy())
    o1, o4 = SurfelOracle("f64", 1), SurfelOracle("f64", 4)
    f1, f4 = o1.forward(*a, s, **kw), o4.forward(*a, s, **kw)
    np.testing.assert_array_equal(f1["color"], f4["color"])
    np.testing.assert_array_equal(f1["point_list"], f4["point_list"])
    g = np.random.default_rng(0)
    gc, ga = g.standard_normal((3, 64, 48)), g.standard_normal((7, 64, 48))
    b1, b4 = o1.backward(f1, gc, ga), o4.backward(f4, gc, ga)
    for k in ("means3D", "scales", "rotations", "opacities", "shs", "means2D"):
        assert U.rel_inf(b4[k], b1[k]) < 1e-12, k


def test_knn_oracle_known_answers(oracle_built):
    """simple_knn.distCUDA2 restatement: unit lattice (every interior point has six neighbours at distance 1 ->
    mean 1), coincident points (distance 0 counts), fewer than four points (inf), and a random set vs numpy."""
    from oracle.gsr_oracle import knn_mean_dist2

    g = np.stack(np.meshgrid(np.arange(5.0), np.arange(5.0), np.arange(5.0), indexing="ij"), -1).reshape(-1, 3)
    d = knn_mean_dist2(g, "f64")
    inner = ((g >= 1) & (g <= 3)).all(1)
    np.testing.assert_allclose(d[inner], 1.0)
    corner = (g == 0).all(1)
    np.testing.assert_allclose(d[corner], 1.0)            # three axis neighbours at distance 1
    pts = np.array([[0, 0, 0], [0, 0, 0], [1, 0, 0], [0, 2, 0], [5, 5, 5.0]])
    d = knn_mean_dist2(pts, "f64")
    np.testing.assert_allclose(d[0], (0 + 1 + 4) / 3)
    np.testing.assert_allclose(d[2], (1 + 1 + 5) / 3)
    assert np.isinf(knn_mean_dist2(pts[:3], "f64")).all() and knn_mean_dist2(np.zeros((0, 3)), "f32").shape == (0,)
    r = np.random.default_rng(0).standard_normal((300, 3))
    d2 = ((r[:, None] - r[None]) ** 2).sum(-1)
    np.fill_diagonal(d2, np.inf)
    np.testing.assert_allclose(knn_mean_dist2(r, "f64", nthreads=4), np.sort(d2, 1)[:, :3].mean(1), rtol=1e-12)
