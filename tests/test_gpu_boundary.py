"""-m gpu: boundary behaviour of the two rasterizer packages that the reference's callers rely on implicitly
(/root/reference/lightning/renderer.py:250-259, renderer_2dgs.py:224-234): an empty Gaussian set differentiates to empty
gradients, views of different sizes render through the multi-view node, and the autograd node notices an in-place update
of its inputs between forward and backward (upstream saves them with save_for_backward)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _settings(H, W, dev, deg=1):
    from generativedensification_amd.camera import orbit_cameras
    from generativedensification_amd.rasterizer import GaussianRasterizationSettings
    cam = orbit_cameras(4, W, H, device=dev)[1]
    return GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(0.375), tanfovy=math.tan(0.375), bg=torch.ones(3, device=dev),
        scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=deg,
        campos=cam.camera_center, prefiltered=False, debug=False)


@pytest.mark.parametrize("surfel", [False, True])
def test_backward_through_an_empty_gaussian_set(surfel):
    """N = 0 through GaussianRasterizer(...) and .backward(): the single-view gdr_backward / gsr_backward used to
    reject the (NULL) empty gradient buffers before looking at N (round-1 advisor finding)."""
    dev = torch.device(DEV)
    rs = _settings(48, 64, dev)
    if surfel:
        from diff_surfel_rasterization import GaussianRasterizer
    else:
        from diff_gaussian_rasterization import GaussianRasterizer
    z = lambda *s: torch.zeros(*s, device=dev, requires_grad=True)
    means, m2d, shs, op = z(0, 3), z(0, 4), z(0, 4, 3), z(0, 1)
    scales, rots = z(0, 2 if surfel else 3), z(0, 4)
    out = GaussianRasterizer(rs)(means3D=means, means2D=m2d, opacities=op, shs=shs, scales=scales, rotations=rots)
    color = out[0]
    assert torch.allclose(color, torch.ones_like(color))       # background only
    sum(o.float().sum() for o in out if o.dtype.is_floating_point).backward()
    torch.cuda.synchronize()
    for t in (means, shs, op, scales, rots):   # autograd may leave the gradient of an empty leaf unset; if set, it is empty
        assert t.grad is None or t.grad.shape == t.shape


def test_render_views_with_mixed_image_sizes_matches_per_view_calls():
    """3DGS render_views / render_views_loss with views of different sizes: one node per size (the surfel twin already
    did this); same images, losses and gradients as the per-view reference call pattern."""
    from generativedensification_amd.camera import orbit_cameras
    from generativedensification_amd.renderer import Renderer
    from generativedensification_amd.synthetic import make_scene, make_targets, view_loss
    dev = torch.device(DEV)
    scene = make_scene(20_000, 5, sh_degree=1, sigma0=(0.01,), device=dev)
    sizes = [(96, 128), (64, 80), (96, 128)]
    cams = [orbit_cameras(3, w, h, device=dev)[j] for j, (h, w) in enumerate(sizes)]
    targets = [make_targets(1, h, w, 7 + j)[0].to(dev) for j, (h, w) in enumerate(sizes)]
    r = Renderer(sh_degree=1)
    r.set_bg_color(torch.ones(3, device=dev))

    def grads_of(fn):
        p = {k: v.clone().requires_grad_(True) for k, v in scene.items()}
        losses = fn(p)
        losses.sum().backward()
        return losses.detach().cpu().numpy(), {k: v.grad.cpu().numpy() for k, v in p.items()}

    def per_view(p):
        return torch.stack([view_loss(r.render_img(c, None, p["centers"], p["shs"], p["opacity"], p["scales"],
                                                   p["rotations"], dev), t) for c, t in zip(cams, targets)])

    def one_call(p):
        outs = r.render_views(cams, None, p["centers"], p["shs"], p["opacity"], p["scales"], p["rotations"], dev)
        assert [tuple(o["image"].shape) for o in outs] == [(h, w, 3) for h, w in sizes]
        return torch.stack([view_loss(o, t) for o, t in zip(outs, targets)])

    def folded(p):
        return r.render_views_loss(cams, None, [t.permute(2, 0, 1).contiguous() for t in targets], p["centers"], p["shs"],
                                   p["opacity"], p["scales"], p["rotations"], dev)

    l0, g0 = grads_of(per_view)
    for fn in (one_call, folded):
        l1, g1 = grads_of(fn)
        np.testing.assert_allclose(l1, l0, rtol=1e-5)
        for k in g0:
            tol = 1e-4 * np.abs(g0[k]) + 1e-6 * np.abs(g0[k]).max()
            assert (np.abs(g1[k] - g0[k]) > tol).mean() < 1e-4, (fn.__name__, k)


@pytest.mark.parametrize("entry", ["rasterizer", "render_views", "surfel"])
def test_inplace_update_between_forward_and_backward_is_an_error(entry):
    """The input tensors are saved with ctx.save_for_backward: K9 recomputes covariance / projection / SH terms from
    them, so a write in between must raise (autograd's version check) rather than return gradients of other values."""
    from generativedensification_amd.synthetic import make_scene
    dev = torch.device(DEV)
    scene = make_scene(2000, 3, sh_degree=1, sigma0=(0.02,), device=dev)
    p = {k: v.requires_grad_(True) for k, v in scene.items()}
    rs = _settings(64, 64, dev)
    if entry == "rasterizer":
        from diff_gaussian_rasterization import GaussianRasterizer
        act = dict(means3D=p["centers"] * 1.0, opacities=torch.sigmoid(p["opacity"]), shs=p["shs"] * 1.0,
                   scales=torch.exp(p["scales"]), rotations=torch.nn.functional.normalize(p["rotations"]))
        out = GaussianRasterizer(rs)(means2D=torch.zeros(2000, 4, device=dev, requires_grad=True), **act)[0]
        victim = act["means3D"]
    elif entry == "surfel":
        from diff_surfel_rasterization import GaussianRasterizer
        act = dict(means3D=p["centers"] * 1.0, opacities=torch.sigmoid(p["opacity"]), shs=p["shs"] * 1.0,
                   scales=torch.exp(p["scales"][:, :2]), rotations=torch.nn.functional.normalize(p["rotations"]))
        out = GaussianRasterizer(rs)(means2D=torch.zeros(2000, 4, device=dev, requires_grad=True), **act)[0]
        victim = act["means3D"]
    else:
        from generativedensification_amd.rasterizer import render_views_raw
        victim = p["centers"] * 1.0
        out = render_views_raw(victim, torch.zeros(2000, 4, device=dev, requires_grad=True), p["shs"], p["opacity"],
                               p["scales"], p["rotations"], [rs, rs])[0][1]
    with torch.no_grad():
        victim.add_(0.01)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        out.sum().backward()


def test_device_topk_of_the_densification_score_matches_torch_topk():
    """gdr_topk_absgrad (radix select) == the mask the reference builds from torch.topk(||grad[:, 2:4]||, k)
    (network.py:876-893): identical sets wherever the k-th score is not tied, correct counts with ties / zeros /
    a candidate mask / k >= N."""
    from generativedensification_amd.rasterizer import topk_absgrad
    dev = torch.device(DEV)
    g = torch.Generator().manual_seed(3)
    for N, k in ((200_000, 12_000), (5_000, 12_000), (70_001, 1), (1000, 999)):
        grad = torch.randn(N, 4, generator=g).to(dev) * torch.rand(N, 1, generator=g).to(dev) ** 4
        score = grad[:, 2:4].norm(dim=-1)
        mask, idx = topk_absgrad(grad, k, return_indices=True)
        kk = min(k, N)
        assert mask.dtype == torch.bool and int(mask.sum()) == kk and idx.shape == (kk,)
        assert set(idx.tolist()) == set(torch.nonzero(mask).flatten().tolist())
        if k < N:
            ref = torch.topk(score, k).indices
            kth = float(score[ref].min())
            sure = score > kth * (1 + 1e-6)                 # (our sqrt(fma) vs torch.norm: an ulp at the threshold)
            assert bool(mask[sure].all()) and not bool(mask[score < kth * (1 - 1e-6)].any())
        else:
            assert bool(mask.all())
    # many exact ties at the threshold (zeros): exactly k selected, every non-zero score among them
    grad = torch.zeros(50_000, 4, device=dev)
    grad[:300, 2] = torch.arange(1, 301, device=dev).float()
    mask = topk_absgrad(grad, 1000)
    assert int(mask.sum()) == 1000 and bool(mask[:300].all())
    # candidates (the reference's grad[mask]): only candidates are selected; k beyond their number selects all of them
    cand = torch.zeros(50_000, dtype=torch.bool, device=dev)
    cand[100:400] = True
    mask = topk_absgrad(grad, 50, candidates=cand)
    assert int(mask.sum()) == 50 and bool(mask[250:300].all()) and not bool(mask[~cand].any())
    mask = topk_absgrad(grad, 12_000, candidates=cand)
    assert bool((mask == cand).all())
    # candidates != None with #candidates <= k < N (the library cannot know the candidate count up front: pass 0 counts)
    mask, idx = topk_absgrad(grad, 300, candidates=cand, return_indices=True)
    assert bool((mask == cand).all()) and sorted(idx.tolist()) == list(range(100, 400))
    mask = topk_absgrad(grad, 400, candidates=cand)
    assert bool((mask == cand).all())
    # NaN scores follow the reference's two branches (network.py:885-890): with >= k candidates torch.topk ranks NaN
    # above every number — they are selected first; with fewer than k the mask is `score >= 0`, false for NaN
    grad = torch.rand(1000, 4, device=dev)
    grad[::7, 2] = float("nan")
    n_nan = len(range(0, 1000, 7))
    mask = topk_absgrad(grad, 200)
    ref = torch.zeros(1000, dtype=torch.bool, device=dev)
    ref[torch.topk(grad[:, 2:4].norm(dim=-1), 200).indices] = True
    assert int(mask.sum()) == 200 and bool(mask[::7].all()) and bool((mask == ref).all())
    mask = topk_absgrad(grad, 100)            # fewer than the NaN count: only NaN scores are selected
    assert int(mask.sum()) == 100 and not bool(mask[~torch.isnan(grad[:, 2])].any())
    mask = topk_absgrad(grad, 5000)           # k > N: `score >= 0`
    assert int(mask.sum()) == 1000 - n_nan and not bool(mask[::7].any())


@pytest.mark.parametrize("surfel", [False, True])
def test_device_sized_binning_equals_the_read_back_and_an_overflow_repeats_the_view(surfel):
    """rasterizer.DEFER_D: the first call of a shape reads the duplicate count back before binning (upstream's flow);
    later calls bin with the count left on the device and buffers carved for a capacity; a capacity that turns out too
    small repeats the view.  All three must give bit-identical lists, images and fused losses, and leave the exact
    count in the state."""
    from generativedensification_amd import rasterizer as R
    from generativedensification_amd.camera import orbit_cameras
    from generativedensification_amd.synthetic import make_scene, make_targets
    dev = torch.device(DEV)
    V, H, W, N = 3, 160, 208, 40_000
    scene = make_scene(N, 23, sh_degree=1, sigma0=(0.01, 0.002), device=dev)
    cams = orbit_cameras(V, W, H, device=dev)
    targets = make_targets(V, H, W, 5).to(dev).permute(0, 3, 1, 2).contiguous()
    if surfel:
        from generativedensification_amd.renderer_2dgs import Renderer
        key = ("surfel",) + R.shape_key(N, H, W, V)       # (the surfel multi-view node keeps its history on the Python side)
    else:
        from generativedensification_amd.renderer import Renderer
        key = None                                        # (3DGS: every entry's history lives in the library)
    sc = scene["scales"][:, :2].contiguous() if surfel else scene["scales"]
    args = (scene["centers"], scene["shs"], scene["opacity"], sc, scene["rotations"], dev)
    if surfel:
        from generativedensification_amd.camera import build_rays
        r = Renderer(sh_degree=1)
        rays = [build_rays(torch.inverse(c.world_view_transform.T.cpu()), 0.75, 0.75, H, W).to(dev) for c in cams]
        views = lambda: r.render_views(cams, rays, None, *args)
        fused = lambda: r.render_views_loss(cams, rays, None, targets, *args)
        single = lambda: r.render_img(cams[1], rays[1], *args)
    else:
        r = Renderer(sh_degree=1, white_background=True)
        r.set_bg_color(torch.ones(3, device=dev))
        views = lambda: r.render_views(cams, None, *args)
        fused = lambda: r.render_views_loss(cams, None, targets, *args)
        single = lambda: r.render_img(cams[1], None, *args)

    # the boundary's own single call (GaussianRasterizer of either package): ONE native call, gdr_forward_view /
    # gsr_forward_view, whose per-shape history lives in the library (gdr_view_history_*)
    r_plain = Renderer(sh_degree=1) if surfel else Renderer(sh_degree=1, white_background=True, fused=False)
    plain = (lambda: r_plain.render_img(cams[1], rays[1], *args)) if surfel else (lambda: r_plain.render_img(cams[1], None, *args))
    from generativedensification_amd import _lib as L
    lib = L.load()

    def run(hint):
        def prep():
            if hint == "none":
                R._D_HINT.clear()
                lib.gdr_view_history_reset()
            elif hint == "small":       # capacity ~4100 entries: every view overflows and is repeated
                if key is not None:
                    R._D_HINT[key] = 1e-4
                lib.gdr_view_history_set(N, H, W, int(surfel), 1e-4)
                lib.gdr_view_history_set(N, H, W, 0, 1e-4)
        res = []
        with torch.no_grad():
            for fn in (views, fused, single, plain):
                prep()
                res.append(fn())
        torch.cuda.synchronize()
        return [o["image"].clone() for o in res[0]], res[1].clone(), res[2]["image"].clone(), res[3]["image"].clone()

    saved, saved_defer = dict(R._D_HINT), R.K.DEFER_D
    R.K.DEFER_D = True
    try:
        img0, loss0, one0, pl0 = run("none")                 # no history: read-back flow
        views()
        assert key is None or (key in R._D_HINT and R._d_capacity(key, N) > R._D_HINT[key] * N > 0)
        assert lib.gdr_view_history_get(N, H, W, int(surfel)) > 1e-2      # (the library's history of the shape)
        img1, loss1, one1, pl1 = run("history")              # device-sized calls
        img2, loss2, one2, pl2 = run("small")
        assert lib.gdr_view_history_get(N, H, W, int(surfel)) > 1e-2    # (the overflow call recorded the real count)
        for imgs, losses, one, pl in ((img1, loss1, one1, pl1), (img2, loss2, one2, pl2)):
            assert all(torch.equal(a, b) for a, b in zip(imgs, img0)) and torch.equal(one, one0) and torch.equal(pl, pl0)
            np.testing.assert_allclose(losses.cpu().numpy(), loss0.cpu().numpy(), rtol=2e-6)   # (atomic order)
        saved_d = R.K.DEFER_D
        R.K.DEFER_D = False          # upstream's flow in every call: the count is read back before anything is sized
        try:
            with torch.no_grad():
                assert torch.equal(plain()["image"], pl0)
        finally:
            R.K.DEFER_D = saved_d
        # the state of a device-sized call holds the exact count and the same sorted lists
        if not surfel:
            sets = [Renderer(sh_degree=1).set_rasterizer(c, device=dev).raster_settings for c in cams]
            res = []
            for hint in (None, 1e-4, "real"):
                if hint is None:
                    lib.gdr_view_history_reset()
                elif hint != "real":
                    lib.gdr_view_history_set(N, H, W, 0, hint)
                with torch.no_grad():
                    states = R._forward_views_impl(scene["centers"], torch.empty(0, 4, device=dev), scene["shs"],
                                                   scene["opacity"], scene["scales"], scene["rotations"], tuple(sets),
                                                   R.RAW_ALL)[4]
                torch.cuda.synchronize()
                assert bool(states[0].bin.d_dev) == (hint == "real")     # read-back / exact repeat / device-sized with a capacity that fits
                res.append([st.tensors() for st in states])
            for other in res[1:]:
                for a, b in zip(res[0], other):
                    assert a["num_rendered"] == b["num_rendered"] > 0
                    for k in ("keys_sorted", "point_list", "ranges", "n_contrib"):
                        assert torch.equal(a[k], b[k]), k
    finally:
        R.K.DEFER_D = saved_defer
        R._D_HINT.clear()
        R._D_HINT.update(saved)


def _report(lib, N, H, W, row=0, words=None):
    import ctypes as C
    buf = (C.c_uint32 * 4)(*(words or (0, 0, 0, 0)))
    assert lib.gdr_view_history_report(N, H, W, 0, row, buf, int(words is not None)) == 0
    return [int(x) for x in buf]


def test_launch_hints_only_size_launches():
    """gdr_binning.stats_out / hint_* (the library's per-shape history, gdr_view_history_report): the binning stage reports
    its tile classes and the next call of the shape sizes the long / medium tile-sort grids and skips the deep-forward launch
    from that.  Wrong hints (an object-like scene after hints that say "no long lists, no deep forward") must give the same
    sorted lists and contributor counts, and the same image up to the deep forward's summation order."""
    from generativedensification_amd import _lib as L
    from generativedensification_amd import rasterizer as R
    from generativedensification_amd.camera import orbit_cameras
    from generativedensification_amd.renderer import Renderer
    from generativedensification_amd.synthetic import make_scene
    dev = torch.device(DEV)
    lib = L.load()
    V, H, W, N = 2, 256, 256, 400_000
    scene = make_scene(N, 29, sh_degree=1, sigma0=(0.004,), device=dev, layout="shell")   # long lists at the silhouette
    cams = orbit_cameras(V, W, H, device=dev)
    sets = [Renderer(sh_degree=1).set_rasterizer(c, device=dev).raster_settings for c in cams]

    def run():
        with torch.no_grad():
            colors, _, _, _, states, _, _ = R._forward_views_impl(scene["centers"], torch.empty(0, 4, device=dev), scene["shs"],
                                                                  scene["opacity"], scene["scales"], scene["rotations"],
                                                                  tuple(sets), R.RAW_ALL)
        torch.cuda.synchronize()
        return colors, [st.tensors() for st in states], [(st.bin.hint_long, st.bin.hint_medium, st.bin.hint_no_deep) for st in states]

    lib.gdr_view_history_reset()
    try:
        run()                                                # first call of the shape: read-back flow, worst-case grids ...
        c0, t0, h0 = run()                                   # ... and the reference result (hints from the first report)
        stats = [_report(lib, N, H, W, v) for v in range(V)]
        assert all(w != 0xFFFFFFFF for st in stats for w in st) and max(st[0] for st in stats) > 0, stats   # lists > 4096 entries exist
        n_long = max(st[0] for st in stats)
        c1, t1, h1 = run()                                   # sized from the report: max over the views + 25 %, floors 16 / 32
        assert all(h[0] == max(16, n_long + n_long // 4 + 1) and h[1] >= 32 for h in h1), (h1, n_long)
        for v in range(V):                                   # "no long / medium lists, no deep forward": all wrong
            _report(lib, N, H, W, v, (0, 0, 0, 0))
        c2, t2, h2 = run()
        # hint_long = -1: the long class (144 KB of LDS per workgroup) is not launched at all; the medium class sorts the
        # lists beyond its capacity through its global-memory bucket pass — same lists
        assert all(h == (-1, 32, 1) for h in h2), h2
        assert max(_report(lib, N, H, W, v)[0] for v in range(V)) > 0       # (and the report is right again)
        for cs, ts in ((c1, t1), (c2, t2)):
            for v in range(V):
                for k in ("keys_sorted", "point_list", "ranges", "n_contrib"):
                    assert torch.equal(ts[v][k], t0[v][k]), k
                assert float((cs[v] - c0[v]).abs().max()) < 2e-5
    finally:
        lib.gdr_view_history_reset()


def test_deep_forward_applies_to_few_busy_tiles_with_long_lists_only():
    """gdr_binning.deep_max_busy / deep_min_mean: the binning stage's report (stats_out[2]) says whether the deep forward
    applied.  An object-like scene with long lists: yes; the same scene with the threshold above its mean list length, or
    with short lists (few Gaussians): no — and the image is the same either way up to the deep forward's summation order."""
    from generativedensification_amd import rasterizer as R
    from generativedensification_amd.camera import orbit_cameras
    from generativedensification_amd.renderer import Renderer
    from generativedensification_amd.synthetic import make_scene
    dev = torch.device(DEV)
    H = W = 256
    cams = orbit_cameras(1, W, H, device=dev)
    sets = [Renderer(sh_degree=1).set_rasterizer(c, device=dev).raster_settings for c in cams]

    from generativedensification_amd import _lib as L
    lib = L.load()

    def run(N, min_mean):
        scene = make_scene(N, 31, sh_degree=1, sigma0=(0.004,), device=dev, layout="shell")
        saved = R.K.DEEP_MIN_MEAN
        R.K.DEEP_MIN_MEAN = min_mean
        lib.gdr_view_history_reset()
        try:
            with torch.no_grad():
                colors, _, _, _, states, _, _ = R._forward_views_impl(scene["centers"], torch.empty(0, 4, device=dev), scene["shs"],
                                                                      scene["opacity"], scene["scales"], scene["rotations"],
                                                                      tuple(sets), R.RAW_ALL)
            torch.cuda.synchronize()
            t = states[0].tensors()
            L_ = (t["ranges"][:, 1] - t["ranges"][:, 0]).long()
            busy = L_[L_ >= 64]
            return colors[0], _report(lib, N, H, W), float(busy.float().mean()), int(busy.numel())
        finally:
            R.K.DEEP_MIN_MEAN = saved
            lib.gdr_view_history_reset()

    c_deep, st_deep, mean_long, busy_long = run(400_000, None)
    assert busy_long <= 768 and mean_long >= 2560, (busy_long, mean_long)    # the premise: few busy tiles, long lists
    assert st_deep[2] == 1 and st_deep[3] == busy_long, st_deep
    c_std, st_std, _, _ = run(400_000, int(mean_long) + 1000)                # threshold above this scene's mean: standard K6
    assert st_std[2] == 0, st_std
    assert float((c_deep - c_std).abs().max()) < 2e-5
    _, st_short, mean_short, busy_short = run(40_000, None)                  # same object, a tenth of the Gaussians: short lists
    assert busy_short <= 768 and mean_short < 2560, (busy_short, mean_short)
    assert st_short[2] == 0, st_short


def test_host_copy_entries_deliver_device_words_to_pinned_memory():
    """gdr_host_copy_begin / gdr_host_copy_wait (include/gdr.h, v13): the pooled-event read-back the Python boundary uses
    for the duplicate count; tickets are reusable after the wait, NULL arguments are refused."""
    import ctypes as C
    from generativedensification_amd import _lib as L
    lib = L.load()
    dev = torch.device(DEV)
    host = torch.zeros(4, dtype=torch.int32).pin_memory()
    seen = set()
    for k in range(5):
        src = torch.arange(4, dtype=torch.int32, device=dev) + 10 * k
        ticket = C.c_void_p()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert lib.gdr_host_copy_begin(host.data_ptr(), src.data_ptr(), 16, stream, C.byref(ticket)) == 0
        assert lib.gdr_host_copy_wait(ticket) == 0
        assert host.tolist() == [10 * k, 10 * k + 1, 10 * k + 2, 10 * k + 3]
        seen.add(ticket.value)
    assert len(seen) == 1                      # the event went back to the pool and was taken again
    assert lib.gdr_host_copy_wait(None) != 0
    assert lib.gdr_host_copy_begin(None, None, 16, None, C.byref(C.c_void_p())) != 0


def test_history_of_a_shape_carries_over_to_a_neighbouring_gaussian_count():
    """A densifying model renders a different N every step: the histories are keyed by the power-of-two bucket of N and
    hold duplicates PER GAUSSIAN, so a call with 25 % more Gaussians than the last one is already device-sized — and
    gives the lists of the read-back flow."""
    from generativedensification_amd import rasterizer as R
    from generativedensification_amd.camera import orbit_cameras
    from generativedensification_amd.renderer import Renderer
    from generativedensification_amd.synthetic import make_scene
    dev = torch.device(DEV)
    V, H, W, N = 2, 160, 208, 50_000
    scene = make_scene(N, 37, sh_degree=1, sigma0=(0.01, 0.002), device=dev)
    cams = orbit_cameras(V, W, H, device=dev)
    sets = [Renderer(sh_degree=1).set_rasterizer(c, device=dev).raster_settings for c in cams]

    def run(n):
        with torch.no_grad():
            colors, _, _, _, states, _, _ = R._forward_views_impl(
                scene["centers"][:n], torch.empty(0, 4, device=dev), scene["shs"][:n], scene["opacity"][:n],
                scene["scales"][:n], scene["rotations"][:n], tuple(sets), R.RAW_ALL)
        torch.cuda.synchronize()
        return colors, states

    from generativedensification_amd import _lib as L
    lib = L.load()
    saved_defer = R.K.DEFER_D
    try:
        R.K.DEFER_D = False
        c_ref, s_ref = run(N)
        R.K.DEFER_D = True
        lib.gdr_view_history_reset()
        _, s0 = run(40_000)
        assert not any(st.bin.d_dev for st in s0)            # first call of the shape: read-back, then exactly sized
        assert lib.gdr_view_history_get(N, H, W, 0) == lib.gdr_view_history_get(40_000, H, W, 0) > 0   # one bucket of N
        c1, s1 = run(N)
        assert all(st.bin.d_dev for st in s1)                # 25 % more Gaussians: device-sized from the ratio
        for v in range(V):
            a, b = s_ref[v].tensors(), s1[v].tensors()
            assert a["num_rendered"] == b["num_rendered"]
            for k in ("keys_sorted", "point_list", "ranges", "n_contrib"):
                assert torch.equal(a[k], b[k]), k
            assert torch.equal(c_ref[v], c1[v])
    finally:
        R.K.DEFER_D = saved_defer
        lib.gdr_view_history_reset()


@pytest.mark.parametrize("surfel", [False, True])
def test_second_backward_through_a_node_clears_its_own_records(surfel):
    """rasterizer._early_records: the gradient records are zero-filled at the end of the forward, on the streams K7 will
    use, and consumed by the first backward; a second backward through the same node (retain_graph) must not find the
    first one's sums in them: .grad after two backwards == 2 x .grad after one."""
    from generativedensification_amd.camera import build_rays, orbit_cameras
    from generativedensification_amd.synthetic import make_scene, make_targets
    dev = torch.device(DEV)
    V, H, W, N = 3, 128, 160, 20_000
    scene = make_scene(N, 43, sh_degree=1, sigma0=(0.01, 0.003), device=dev)
    if surfel:
        scene["scales"] = scene["scales"][:, :2].contiguous()
    cams = orbit_cameras(V, W, H, device=dev)
    targets = make_targets(V, H, W, 3).to(dev).permute(0, 3, 1, 2).contiguous()
    if surfel:
        from generativedensification_amd.renderer_2dgs import Renderer
        r = Renderer(sh_degree=1)
        rays = [build_rays(torch.inverse(c.world_view_transform.T.cpu()), 0.75, 0.75, H, W).to(dev) for c in cams]
    else:
        from generativedensification_amd.renderer import Renderer
        r = Renderer(sh_degree=1, white_background=True)
        r.set_bg_color(torch.ones(3, device=dev))
    for fused_loss in (True, False):
        grads = []
        for times in (1, 2):
            p = {k: v.clone().requires_grad_(True) for k, v in scene.items()}
            a = (p["centers"], p["shs"], p["opacity"], p["scales"], p["rotations"], dev)
            if fused_loss:
                lv = r.render_views_loss(cams, rays, None, targets, *a) if surfel else r.render_views_loss(cams, None, targets, *a)
            else:
                outs = r.render_views(cams, rays, None, *a) if surfel else r.render_views(cams, None, *a)
                lv = torch.stack([((o["image"].permute(2, 0, 1) - targets[j]) ** 2).mean() for j, o in enumerate(outs)])
            loss = lv.sum()
            for t in range(times):
                loss.backward(retain_graph=t + 1 < times)
            torch.cuda.synchronize()
            grads.append({k: v.grad.cpu().numpy() for k, v in p.items()})
        for k in grads[0]:
            ref = 2.0 * grads[0][k]
            tol = 1e-4 * np.abs(ref) + 1e-6 * np.abs(ref).max()
            assert (np.abs(grads[1][k] - ref) > tol).mean() < 1e-4, (fused_loss, k)


def test_k7_variant_is_measured_per_shape_and_can_be_pinned():
    """include/gdr.h gdr_k7_tune_override / gdr_k7_tune_get (v15): four backward launches of a scene shape (after its first 8)
    are timed (rows / pairs / rows / pairs), then one variant serves; an override pins it; the gradients do not depend on any
    of it beyond the order of the float atomics."""
    import ctypes as C

    import util as U
    from generativedensification_amd import _lib as L

    lib = L.load()
    case = U.make_case(20_000, 96, 112, 23, deg=1, sigma0=(0.03, 0.01))
    grads = U.rand_grads(case)
    N, H, W = 20_000, 96, 112
    chosen, us0, us1 = C.c_int32(-7), C.c_float(0), C.c_float(0)
    lib.gdr_view_history_reset()
    try:
        lib.gdr_k7_tune_override(-1)
        ref = None
        for i in range(16):     # (the round of four timed launches starts after the shape's first 8)
            _, g = U.run_hip(case, grads)
            ref = ref or g
            for k in ("means3D", "shs", "opacities"):
                assert U.rel_inf(g[k], ref[k]) < 5e-5, (i, k)
        torch.cuda.synchronize()
        assert lib.gdr_k7_tune_get(N, H, W, 1, 0, C.byref(chosen), C.byref(us0), C.byref(us1)) == 0
        assert chosen.value in (0, 1) and us0.value > 0 and us1.value > 0, (chosen.value, us0.value, us1.value)
        assert lib.gdr_k7_tune_get(N, H + 16, W, 1, 0, C.byref(chosen), C.byref(us0), C.byref(us1)) != 0    # no launch of that shape
        for mode in (0, 1):
            lib.gdr_k7_tune_override(mode)
            _, g = U.run_hip(case, grads)
            for k in ("means3D", "shs", "opacities"):
                assert U.rel_inf(g[k], ref[k]) < 5e-5, (mode, k)
    finally:
        lib.gdr_k7_tune_override(-1)


def test_k7_choice_gets_one_confirmation_round_that_can_overturn_a_wrong_first_pick():
    """include/gdr.h v17 (round-5 verdict next #8): launches 64..67 of a shape time both K7 kernels once more; the first pick is
    overturned iff it loses that round by > 5 %, and the choice is final afterwards.  Both first picks are forced in turn
    (gdr_k7_tune_force_first) so that whichever kernel is slower on this box is, once, the 'wrong first pick'; the outcome
    must follow the rule applied to the times the library reports, and the slower-by-5-% kernel never survives."""
    import ctypes as C

    import util as U
    from generativedensification_amd import _lib as L

    lib = L.load()
    N, H, W = 20_000, 96, 112
    case = U.make_case(N, H, W, 23, deg=1, sigma0=(0.03, 0.01))
    grads = U.rand_grads(case)
    chosen, rounds, us4 = C.c_int32(-7), C.c_int32(-7), (C.c_float * 4)()
    finals = []
    try:
        lib.gdr_k7_tune_override(-1)
        for first in (0, 1):
            lib.gdr_view_history_reset()
            for _ in range(13):
                U.run_hip(case, grads)
            torch.cuda.synchronize()
            U.run_hip(case, grads)                      # (harvests the first round)
            assert lib.gdr_k7_tune_get_rounds(N, H, W, 1, 0, C.byref(chosen), C.byref(rounds), us4) == 0
            assert rounds.value == 1 and us4[0] > 0 and us4[1] > 0 and us4[2] == 0
            assert lib.gdr_k7_tune_force_first(N, H, W, 1, 0, first) == 0
            for _ in range(60):
                U.run_hip(case, grads)
            torch.cuda.synchronize()
            U.run_hip(case, grads)                      # (harvests the confirmation round)
            assert lib.gdr_k7_tune_get_rounds(N, H, W, 1, 0, C.byref(chosen), C.byref(rounds), us4) == 0
            assert rounds.value == 2 and us4[2] > 0 and us4[3] > 0, (rounds.value, list(us4))
            own, other = us4[2 + first], us4[3 - first]
            want = (1 - first) if other < 0.95 * own else first
            assert chosen.value == want, (first, chosen.value, list(us4))
            finals.append((chosen.value, us4[2], us4[3]))
            assert lib.gdr_k7_tune_force_first(N, H, W, 1, 0, 1 - first) != 0      # final: no further change
        for ch, rows_us, pairs_us in finals:            # a kernel that lost its confirmation round by > 5 % never serves
            assert not (ch == 0 and pairs_us < 0.95 * rows_us) and not (ch == 1 and rows_us < 0.95 * pairs_us)
        print("[k7 confirmation] (chosen, rows us, pairs us) after forcing rows / pairs first:", finals)
    finally:
        lib.gdr_k7_tune_override(-1)
        lib.gdr_view_history_reset()


def test_gradient_sinks_k9_writes_where_the_collective_needs_it():
    """Round 5 (multiview.prepare_grad_sinks / rasterizer.register_grad_sink): with the slices of a packed buffer registered as
    the gradient sinks of the leaves, the multi-view node's K9 writes there directly — `.grad` IS the slice (no copy), bit for
    bit what the node returns without sinks; an existing `.grad`, a second node in the same pass and a non-leaf input all
    take ordinary buffers."""
    from generativedensification_amd import rasterizer as R
    from generativedensification_amd.camera import orbit_cameras
    from generativedensification_amd.renderer import Renderer
    from generativedensification_amd.synthetic import make_scene, make_targets
    dev = torch.device("cuda:0")
    N, H, W, V, deg = 20_000, 96, 128, 3, 1
    sc = make_scene(N, 11, sh_degree=deg, sigma0=(0.0052, 0.02), device=dev)
    cams = orbit_cameras(V, W, H, device=dev)
    tg = make_targets(V, H, W, 11).to(dev).permute(0, 3, 1, 2).contiguous()
    r = Renderer(sh_degree=deg)
    r.set_bg_color(torch.ones(3, device=dev))
    keys = ("centers", "shs", "opacity", "scales", "rotations")

    def run(leaves, twice=False):
        lv = r.render_views_loss(cams, None, tg, *[leaves[k] for k in keys], dev)
        if twice:
            lv = lv + r.render_views_loss(cams, None, tg, *[leaves[k] for k in keys], dev)
        lv.sum().backward()
        return {k: leaves[k].grad for k in keys}

    def close(a, b):      # (two runs differ in the order of K7's float atomics: the per-element bar of the parity tests)
        import util as U
        out, _, maxn = U.elem_stats(a.cpu().numpy(), b.cpu().numpy())
        return out < U.MAX_OUTSIDE and maxn < 1e-4

    R.unregister_grad_sinks()
    ref = {k: v.clone() for k, v in run({k: sc[k].clone().requires_grad_(True) for k in keys}).items()}
    leaves = {k: sc[k].clone().requires_grad_(True) for k in keys}
    flat = torch.full((sum(v.numel() for v in leaves.values()),), float("nan"), device=dev)
    sinks, off = {}, 0
    for k in keys:
        sinks[k] = flat[off: off + leaves[k].numel()].view(leaves[k].shape)
        off += leaves[k].numel()
        R.register_grad_sink(leaves[k], sinks[k])
    try:
        g = run(leaves)
        for k in keys:
            assert g[k].data_ptr() == sinks[k].data_ptr(), k                 # adopted: the gradient lives in the buffer
            assert close(g[k], ref[k]), k                                    # and is what the node computes anyway
        assert not torch.isnan(flat).any()
        g2 = run(leaves)                                                     # .grad exists: accumulated in place, never aliased
        for k in keys:
            assert g2[k].data_ptr() == sinks[k].data_ptr() and close(g2[k], 2 * ref[k]), k
        for v in leaves.values():
            v.grad = None
        g3 = run(leaves, twice=True)                                         # two nodes, one pass: one takes the sinks
        for k in keys:
            assert close(g3[k], 2 * ref[k]), k
        mid = {k: v * 1.0 for k, v in leaves.items()}                        # non-leaf inputs: no sink applies
        for v in leaves.values():
            v.grad = None
        flat.fill_(float("nan"))
        run_mid = r.render_views_loss(cams, None, tg, *[mid[k] for k in keys], dev)
        run_mid.sum().backward()
        assert torch.isnan(flat).all()
        for k in keys:
            assert close(leaves[k].grad, ref[k]), k
    finally:
        R.unregister_grad_sinks()
