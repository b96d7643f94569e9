"""-m gpu: render groups (generativedensification_amd/viewgroup.py) — the fast path of the UNCHANGED caller.

The reference renders the views of one Gaussian set one `render_img` at a time and back-propagates once
(/root/reference/lightning/network.py:827-838, 848-856, 964-972; renderer.py:225-259).  Grouped, those calls share ONE
preprocess-backward; the results must be what independent calls give: images bit for bit (the forward is the same code),
leaf gradients within the per-element bar (only the order of fp32 sums over the views changes)."""
import math

import numpy as np
import pytest
import torch

import util as U

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(V=4, n=20_000, h=128, w=160, deg=1, B=2, seed=5):
    from generativedensification_amd.camera import orbit_cameras
    from generativedensification_amd.rasterizer import GaussianRasterizationSettings
    from generativedensification_amd.synthetic import make_scene, make_targets
    dev = torch.device(DEV)
    scenes = [make_scene(n, seed + b, sh_degree=deg, sigma0=(0.0052, 0.02), device=dev) for b in range(B)]
    base = {k: torch.stack([sc[k] for sc in scenes]) for k in scenes[0]}       # (B, N, ...) like the decoder's outputs
    cams = orbit_cameras(V, w, h, device=dev)
    three = ([1.0, 1.0, 1.0], [0.5, 0.5, 0.5], [0.0, 0.0, 0.0])
    sets = [GaussianRasterizationSettings(
        image_height=h, image_width=w, tanfovx=math.tan(0.375), tanfovy=math.tan(0.375),
        bg=torch.tensor(three[j % 3], device=dev), scale_modifier=1.0, viewmatrix=c.world_view_transform,
        projmatrix=c.full_proj_transform, sh_degree=deg, campos=c.camera_center, prefiltered=False, debug=False)
        for j, c in enumerate(cams)]
    tg = make_targets(V, h, w, seed).to(dev).permute(0, 3, 1, 2)
    return dev, base, sets, tg, n


def _reference_loop(leaves, sets, tg, n, dev, i=1, carriers=None):
    """What network.py + renderer.py do for sample i: per view new slices, new activations, a new carrier, one call."""
    import diff_gaussian_rasterization as D
    centers = leaves["centers"][i]                      # (taken once, network.py:821)
    imgs, losses, ssps = [], [], []
    for j, rs in enumerate(sets):
        ssp = (torch.zeros(n, 4, device=dev, requires_grad=True) + 0) if carriers is None else carriers[j]
        ssp.retain_grad()
        color, radii, depth, alpha = D.GaussianRasterizer(rs)(
            means3D=centers, means2D=ssp, shs=leaves["shs"][i], opacities=torch.sigmoid(leaves["opacity"][i]),
            scales=torch.exp(leaves["scales"][i]), rotations=torch.nn.functional.normalize(leaves["rotations"][i]))
        imgs.append(torch.cat([color, depth, alpha]))
        losses.append(((color.clamp(0, 1) - tg[j]) ** 2).mean() + 0.1 * depth.mean() + 0.1 * alpha.mean())
        ssps.append(ssp)
    return imgs, losses, ssps


def _run(grouped, base, sets, tg, n, dev, per_view_backward=False):
    from generativedensification_amd import _lib as L
    from generativedensification_amd import viewgroup as G
    saved = G.GROUP_VIEWS
    G.GROUP_VIEWS = grouped
    try:
        leaves = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        L.profile_enable(True)
        L.profile_collect(reset=True)
        imgs, losses, ssps = _reference_loop(leaves, sets, tg, n, dev)
        if per_view_backward:
            for l in losses:
                l.backward(retain_graph=True)
        else:
            sum(losses).backward()
        torch.cuda.synchronize()
        prof = L.profile_collect(reset=True)
        L.profile_enable(False)
    finally:
        G.GROUP_VIEWS = saved
    return ([x.detach().cpu().numpy() for x in imgs], {k: v.grad.cpu().numpy() for k, v in leaves.items()},
            [s.grad.cpu().numpy() for s in ssps], prof)


@pytest.mark.parametrize("per_view_backward", [False, True])
def test_grouped_calls_equal_independent_calls(per_view_backward):
    dev, base, sets, tg, n = _setup()
    i0, g0, m0, p0 = _run(False, base, sets, tg, n, dev, per_view_backward)
    i1, g1, m1, p1 = _run(True, base, sets, tg, n, dev, per_view_backward)
    V = len(sets)
    assert p0["preprocess_bwd"][1] == V                                   # independent: one K8+K9 per view
    assert p1["preprocess_bwd"][1] == (V if per_view_backward else 1)     # grouped: one per backward pass
    assert p1["render_bwd"][1] == V and p1["preprocess_fwd"][1] == V
    for a, b in zip(i1, i0):
        np.testing.assert_array_equal(a, b)
    for k in g0:
        out, worst, maxn = U.elem_stats(g1[k], g0[k])
        assert out < U.MAX_OUTSIDE and maxn < 1e-4, (k, out, worst, maxn)
        assert np.abs(g0[k][0]).max() == 0 and np.abs(g1[k][0]).max() == 0      # sample 0 was never rendered
    for a, b in zip(m1, m0):      # every call's own (N,4) carrier gradient: K7's records, no reordering at all
        out, worst, maxn = U.elem_stats(a, b)
        assert a.shape == (n, 4) and out < U.MAX_OUTSIDE and maxn < 1e-4 and (a[:, 2:] >= 0).all()


def test_vjp_pass_then_main_pass_and_a_second_gaussian_set():
    """network.py's sequence on one sample: coarse renders, `vjp` w.r.t. a shared carrier through renders of the SAME set
    (network.py:843-872: the hub is not part of that pass, its K7 results must not leak into the next), renders of a
    DIFFERENT set (another group), one backward through everything."""
    from torch.autograd.functional import vjp
    from generativedensification_amd import viewgroup as G
    dev, base, sets, tg, n = _setup(V=3)

    def run(grouped):
        saved = G.GROUP_VIEWS
        G.GROUP_VIEWS = grouped
        try:
            leaves = {k: v.clone().requires_grad_(True) for k, v in base.items()}
            _, losses, _ = _reference_loop(leaves, sets, tg, n, dev, i=1)

            def fn(ssp):
                _, l2, _ = _reference_loop(leaves, sets[:2], tg, n, dev, i=1, carriers=[ssp, ssp])
                return sum(l2)
            with torch.no_grad():
                val, grad = vjp(fn, torch.zeros(n, 4, device=dev))
            _, losses_b, _ = _reference_loop(leaves, sets, tg, n, dev, i=0)
            (sum(losses) + sum(losses_b)).backward()
            return grad.cpu().numpy(), {k: v.grad.cpu().numpy() for k, v in leaves.items()}
        finally:
            G.GROUP_VIEWS = saved

    a0, g0 = run(False)
    a1, g1 = run(True)
    out, _, maxn = U.elem_stats(a1, a0)
    assert out < U.MAX_OUTSIDE and maxn < 1e-4 and np.abs(a0[:, 2:]).max() > 0
    for k in g0:
        out, worst, maxn = U.elem_stats(g1[k], g0[k])
        assert out < U.MAX_OUTSIDE and maxn < 1e-4, (k, out, worst, maxn)
        assert np.abs(g0[k][0]).max() > 0 and np.abs(g0[k][1]).max() > 0


def test_values_that_changed_behind_autograds_back_fall_back_to_an_independent_node():
    """Equal provenance, different values: a source edited in place under no_grad between two calls.  Views of the edited
    tensor carry its version counter (a new group, correct gradients); the activated copies do not — the device-side
    comparison next to K1 catches them, and the offending call is rendered as an ordinary independent node from ITS OWN
    tensors, exactly what the reference's per-call nodes do (round-3 advisor finding: it used to raise).  The whole
    sequence — images and leaf gradients — must equal the ungrouped run."""
    import diff_gaussian_rasterization as D
    from generativedensification_amd import viewgroup as G
    dev, base, sets, tg, n = _setup(V=3, B=1)

    def run(grouped):
        saved = G.GROUP_VIEWS
        G.GROUP_VIEWS = grouped
        try:
            leaves = {k: v.clone().requires_grad_(True) for k, v in base.items()}
            mid = {k: v * 1.0 for k, v in leaves.items()}        # non-leaf sources, like a decoder's outputs

            def call(rs):
                ssp = torch.zeros(n, 4, device=dev, requires_grad=True)
                return D.GaussianRasterizer(rs)(means3D=mid["centers"][0], means2D=ssp, shs=mid["shs"][0],
                                                opacities=torch.sigmoid(mid["opacity"][0]), scales=torch.exp(mid["scales"][0]),
                                                rotations=torch.nn.functional.normalize(mid["rotations"][0]))[0]
            a = call(sets[0])
            b = call(sets[1])
            with torch.no_grad():
                mid["opacity"].add_(0.5)      # sigmoid's backward only needs its OUTPUT: autograd accepts the edit
            c = call(sets[2])                 # same provenance as a and b, other values
            (a.mean() + 2.0 * b.mean() + 3.0 * c.mean()).backward()
            return [x.detach().cpu().numpy() for x in (a, b, c)], {k: v.grad.cpu().numpy() for k, v in leaves.items()}
        finally:
            G.GROUP_VIEWS = saved

    i0, g0 = run(False)
    i1, g1 = run(True)
    for x, y in zip(i1, i0):
        np.testing.assert_array_equal(x, y)
    assert np.abs(i0[2] - i0[1]).max() > 1e-3          # (the edit is visible in the third render)
    for k in g0:
        out, worst, maxn = U.elem_stats(g1[k], g0[k])
        assert out < U.MAX_OUTSIDE and maxn < 1e-4, (k, out, worst, maxn)


def test_watched_activation_tensors_are_never_grouped_and_single_view_passes_pause_grouping():
    """What a caller could observe of a group is ruled out up front: a call whose activation
    tensor carries a hook or retain_grad() is an ordinary node (the later calls' activation tensors of a group receive no
    gradient); and a caller that back-propagates after EVERY view (no gain from groups, only their bookkeeping) is left
    alone after two such passes until it renders several views per pass again."""
    import diff_gaussian_rasterization as D
    from generativedensification_amd import _lib as L
    from generativedensification_amd import viewgroup as G
    dev, base, sets, tg, n = _setup(V=2, B=1)
    leaves = {k: v.clone().requires_grad_(True) for k, v in base.items()}

    def call(rs, watch=None):
        op = torch.sigmoid(leaves["opacity"][0])
        if watch == "retain":
            op.retain_grad()
        elif watch == "hook":
            op.register_hook(lambda g: seen.append(float(g.abs().sum())))
        ssp = torch.zeros(n, 4, device=dev, requires_grad=True)
        out = D.GaussianRasterizer(rs)(means3D=leaves["centers"][0], means2D=ssp, shs=leaves["shs"][0], opacities=op,
                                       scales=torch.exp(leaves["scales"][0]),
                                       rotations=torch.nn.functional.normalize(leaves["rotations"][0]))[0]
        return out, op

    def k9_launches(fn):
        L.profile_enable(True)
        L.profile_collect(reset=True)
        fn()
        torch.cuda.synchronize()
        prof = L.profile_collect(reset=True)
        L.profile_enable(False)
        return prof["preprocess_bwd"][1]

    seen: list = []
    G.pace().solo_passes = 0

    def watched():
        (a, _), (b, op_b) = call(sets[0]), call(sets[1], "retain")
        (c, _) = call(sets[0], "hook")
        (a.mean() + b.mean() + c.mean()).backward()
        assert op_b.grad is not None and float(op_b.grad.abs().sum()) > 0
    assert k9_launches(watched) == 3 and len(seen) == 1 and seen[0] > 0      # three independent nodes, the hook fired

    def two_views():
        (a, _), (b, _) = call(sets[0]), call(sets[1])
        (a.mean() + b.mean()).backward()
    assert k9_launches(two_views) == 1                                       # a group

    def solo():
        call(sets[0])[0].mean().backward()
    for _ in range(2):
        solo()
    assert G.pace().solo_passes >= 2
    assert k9_launches(two_views) == 2      # paused: this pass still runs as independent nodes ...
    assert G.pace().solo_passes == 0
    assert k9_launches(two_views) == 1      # ... and grouping is back with the next one


@pytest.mark.parametrize("deg", [1, 3])
def test_surfel_calls_of_one_set_share_one_preprocess_backward(deg):
    """Render groups for `diff_surfel_rasterization` (round 4; /root/reference/lightning/renderer_2dgs.py:224-234 called
    per view): images and allmaps bit for bit, every call's own carrier gradient, leaf gradients within the per-element
    bar of the surfel path, ONE K9s launch for the pass instead of one per view."""
    import diff_surfel_rasterization as DS
    from generativedensification_amd import _lib as L
    from generativedensification_amd import viewgroup as G
    dev, base, sets, tg, n = _setup(V=4, deg=deg, B=2, seed=9)
    base = dict(base)
    base["scales"] = base["scales"][..., :2].contiguous()
    gmap = torch.randn(4, 7, sets[0].image_height, sets[0].image_width, device=dev) * 0.1
    gmap[:, 6] *= 0.01

    def run(grouped):
        saved = G.GROUP_VIEWS
        G.GROUP_VIEWS = grouped
        G.pace().solo_passes = 0
        try:
            leaves = {k: v.clone().requires_grad_(True) for k, v in base.items()}
            L.profile_enable(True)
            L.profile_collect(reset=True)
            outs, ssps, loss = [], [], 0.0
            for j, rs in enumerate(sets):
                ssp = torch.zeros(n, 4, device=dev, requires_grad=True)
                color, radii, allmap = DS.GaussianRasterizer(rs)(
                    means3D=leaves["centers"][1], means2D=ssp, shs=leaves["shs"][1],
                    opacities=torch.sigmoid(leaves["opacity"][1]), scales=torch.exp(leaves["scales"][1]),
                    rotations=torch.nn.functional.normalize(leaves["rotations"][1]))
                outs.append(torch.cat([color, allmap]))
                ssps.append(ssp)
                loss = loss + ((color.clamp(0, 1) - tg[j]) ** 2).mean() + (allmap * gmap[j]).mean()
            loss.backward()
            torch.cuda.synchronize()
            prof = L.profile_collect(reset=True)
            L.profile_enable(False)
            return ([o.detach().cpu().numpy() for o in outs], {k: v.grad.cpu().numpy() for k, v in leaves.items()},
                    [s.grad.cpu().numpy() for s in ssps], prof)
        finally:
            G.GROUP_VIEWS = saved

    i0, g0, m0, p0 = run(False)
    i1, g1, m1, p1 = run(True)
    assert p0["preprocess_bwd"][1] == 4 and p1["preprocess_bwd"][1] == 1 and p1["render_bwd"][1] == 4
    for a, b in zip(i1, i0):
        np.testing.assert_array_equal(a, b)
    for k in g0:     # only the order of the fp32 sums over the views changes
        out, worst, maxn = U.elem_stats(g1[k], g0[k], 1e-4, U.SURFEL_ATOL_REL)
        assert out < U.MAX_OUTSIDE and maxn < 2e-4, (k, out, worst, maxn)
        assert np.abs(g0[k][0]).max() == 0 and np.abs(g1[k][0]).max() == 0
    for a, b in zip(m1, m0):
        out, worst, maxn = U.elem_stats(a, b, 1e-4, U.SURFEL_ATOL_REL)
        assert a.shape == (n, 4) and out < U.MAX_OUTSIDE and maxn < 2e-4



# ---- a view rendered twice: the forward the vjp pass repeats (network.py:827-838 vs 848-856) -----------------------------
def _fresh_settings(rs, bg=None, view=None):
    """The same 12 fields in NEW device tensors (MiniCam and `bg.to(device)` rebuild them per call in the reference)."""
    return rs._replace(bg=(rs.bg if bg is None else bg).clone(), viewmatrix=(rs.viewmatrix if view is None else view).clone(),
                       projmatrix=rs.projmatrix.clone(), campos=rs.campos.clone())


def _sample_sequence(leaves, sets, vjp_sets, tg, n, dev):
    """network.py's sequence on one sample: V coarse renders; `vjp` of the image MSE w.r.t. ONE shared carrier through renders
    of the same Gaussians with `vjp_sets`; one backward through the coarse renders."""
    from torch.autograd.functional import vjp
    imgs, losses, _ = _reference_loop(leaves, sets, tg, n, dev, i=1)
    seen = []

    def fn(ssp):
        im2, _, _ = _reference_loop(leaves, vjp_sets, tg, n, dev, i=1, carriers=[ssp] * len(vjp_sets))
        seen.extend(x.detach() for x in im2)
        return sum(((x[:3].clamp(0, 1) - tg[j]) ** 2).mean() for j, x in enumerate(im2))
    val, grad = vjp(fn, torch.zeros(n, 4, device=dev))
    sum(losses).backward()
    return ([x.detach().cpu().numpy() for x in imgs], [x.cpu().numpy() for x in seen], float(val), grad.cpu().numpy(),
            {k: v.grad.cpu().numpy() for k, v in leaves.items()})


def _profiled(fn):
    from generativedensification_amd import _lib as L
    L.profile_enable(True)
    L.profile_collect(reset=True)
    out = fn()
    torch.cuda.synchronize()
    prof = L.profile_collect(reset=True)
    L.profile_enable(False)
    return out, prof


def test_a_view_rendered_twice_runs_its_forward_once():
    """The vjp pass of network.py:848-856 renders the first n_views_sel views of the SAME coarse Gaussians with the SAME c2w
    and bg_color that :827-838 rendered moments earlier (new MiniCam / bg tensors, equal values).  The render group hands
    out the first forward's results: V K1 launches instead of V + 2, the images bit for bit, the carrier gradient and the
    leaf gradients as without the reuse."""
    from generativedensification_amd import viewgroup as G
    dev, base, sets, tg, n = _setup(V=4)
    G._REUSE_HIST.clear()

    def run(reuse):
        saved = G.REUSE_FORWARD
        G.REUSE_FORWARD = reuse
        G._REUSE_STATS.update(probes=0, hits=0)
        try:
            leaves = {k: v.clone().requires_grad_(True) for k, v in base.items()}
            vjp_sets = [_fresh_settings(rs) for rs in sets[:2]]
            return _profiled(lambda: _sample_sequence(leaves, sets, vjp_sets, tg, n, dev)), dict(G._REUSE_STATS)
        finally:
            G.REUSE_FORWARD = saved

    (r0, p0), s0 = run(False)
    (r1, p1), s1 = run(True)
    assert s0 == dict(probes=0, hits=0) and s1["hits"] == 2 and s1["probes"] == 5      # first time: every later call probes
    assert p0["preprocess_fwd"][1] == 6 and p1["preprocess_fwd"][1] == 4
    assert p0["render_fwd"][1] == 6 and p1["render_fwd"][1] == 4
    for a, b in zip(r1[0] + r1[1], r0[0] + r0[1]):
        np.testing.assert_array_equal(a, b)
    for j in range(2):
        np.testing.assert_array_equal(r1[1][j], r1[0][j])         # the repeated views ARE the first renders
    assert r1[2] == r0[2]
    out, _, maxn = U.elem_stats(r1[3], r0[3])
    assert out < U.MAX_OUTSIDE and maxn < 1e-4 and np.abs(r0[3][:, 2:]).max() > 0
    for k in r0[4]:
        out, worst, maxn = U.elem_stats(r1[4][k], r0[4][k])
        assert out < U.MAX_OUTSIDE and maxn < 1e-4, (k, out, worst, maxn)
    # the second step of the same shape has learned which call indices repeat a view: only those probe
    (r2, p2), s2 = run(True)
    assert s2 == dict(probes=2, hits=2) and p2["preprocess_fwd"][1] == 4
    for a, b in zip(r2[0] + r2[1], r0[0] + r0[1]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("what", ["bg", "c2w", "inplace_same_tensor"])
def test_a_changed_setting_between_the_passes_takes_the_ordinary_path(what):
    """Another bg colour, another camera, a settings tensor written in place between the calls (same memory: no device
    comparison could tell): no reuse, results equal to the run without it."""
    from generativedensification_amd import viewgroup as G
    dev, base, sets, tg, n = _setup(V=3)
    G._REUSE_HIST.clear()

    def run(reuse):
        saved = G.REUSE_FORWARD
        G.REUSE_FORWARD = reuse
        G._REUSE_STATS.update(probes=0, hits=0)
        try:
            leaves = {k: v.clone().requires_grad_(True) for k, v in base.items()}
            my = [_fresh_settings(rs) for rs in sets]
            imgs, losses, _ = _reference_loop(leaves, my, tg, n, dev, i=1)
            if what == "bg":
                again = _fresh_settings(my[0], bg=1.0 - my[0].bg)
            elif what == "c2w":
                again = _fresh_settings(my[0], view=my[0].viewmatrix + 1e-3)
            else:
                my[0].bg.mul_(0.5)
                again = my[0]
            im2, l2, _ = _reference_loop(leaves, [again], tg, n, dev, i=1)
            (sum(losses) + l2[0]).backward()
            return im2[0].detach().cpu().numpy(), {k: v.grad.cpu().numpy() for k, v in leaves.items()}, dict(G._REUSE_STATS)
        finally:
            G.REUSE_FORWARD = saved

    i0, g0, _ = run(False)
    i1, g1, st = run(True)
    assert st["hits"] == 0 and st["probes"] >= 1
    np.testing.assert_array_equal(i1, i0)
    for k in g0:
        out, worst, maxn = U.elem_stats(g1[k], g0[k])
        assert out < U.MAX_OUTSIDE and maxn < 1e-4, (k, out, worst, maxn)


def test_five_recomputed_inputs_per_call_still_group():
    """Round-4 advisor finding: a caller that recomputes ALL five inputs per call (means3D and shs through whitelisted ops as
    well) hands the forward five same_as pairs; the struct held four and the call crashed with an IndexError."""
    import diff_gaussian_rasterization as D
    dev, base, sets, tg, n = _setup(V=3, B=1)

    def run(grouped):
        from generativedensification_amd import viewgroup as G
        saved = G.GROUP_VIEWS
        G.GROUP_VIEWS = grouped
        try:
            leaves = {k: v.clone().requires_grad_(True) for k, v in base.items()}
            loss = 0.0
            for j, rs in enumerate(sets):
                ssp = torch.zeros(n, 4, device=dev, requires_grad=True)
                color = D.GaussianRasterizer(rs)(
                    means3D=torch.sigmoid(leaves["centers"][0]), means2D=ssp, shs=torch.sigmoid(leaves["shs"][0]),
                    opacities=torch.sigmoid(leaves["opacity"][0]), scales=torch.exp(leaves["scales"][0]),
                    rotations=torch.nn.functional.normalize(leaves["rotations"][0]))[0]
                loss = loss + ((color.clamp(0, 1) - tg[j]) ** 2).mean()
            (_, prof) = _profiled(lambda: loss.backward())
            return {k: v.grad.cpu().numpy() for k, v in leaves.items()}, prof
        finally:
            G.GROUP_VIEWS = saved
    g0, p0 = run(False)
    g1, p1 = run(True)
    assert p0["preprocess_bwd"][1] == 3 and p1["preprocess_bwd"][1] == 1
    for k in g0:
        out, worst, maxn = U.elem_stats(g1[k], g0[k])
        assert out < U.MAX_OUTSIDE and maxn < 1e-4, (k, out, worst, maxn)


def test_an_output_edited_in_place_is_not_handed_out_again():
    """The cache holds aliases of what the first call returned; a caller that edits such an output in place (none of the
    reference's does) must not see the edit in a later call's result: the version counter rules the entry out."""
    import diff_gaussian_rasterization as D
    from generativedensification_amd import viewgroup as G
    dev, base, sets, tg, n = _setup(V=2, B=1)
    G._REUSE_HIST.clear()
    leaves = {k: v.clone().requires_grad_(True) for k, v in base.items()}

    def call(rs):
        ssp = torch.zeros(n, 4, device=dev, requires_grad=True)
        return D.GaussianRasterizer(rs)(means3D=leaves["centers"][0], means2D=ssp, shs=leaves["shs"][0],
                                        opacities=torch.sigmoid(leaves["opacity"][0]), scales=torch.exp(leaves["scales"][0]),
                                        rotations=torch.nn.functional.normalize(leaves["rotations"][0]))
    a = call(sets[0])
    b = call(_fresh_settings(sets[0]))                 # a hit: equal to a, in memory of its own
    assert G._REUSE_STATS["hits"] >= 1 and b[0].data_ptr() != a[0].data_ptr()
    np.testing.assert_array_equal(a[0].detach().cpu().numpy(), b[0].detach().cpu().numpy())
    ref = a[0].detach().clone()
    with torch.no_grad():
        a[3].mul_(0.5)                                 # alpha of the first call edited in place
    hits = G._REUSE_STATS["hits"]
    c = call(_fresh_settings(sets[0]))
    np.testing.assert_array_equal(c[0].detach().cpu().numpy(), ref.cpu().numpy())
    np.testing.assert_array_equal(c[3].detach().cpu().numpy(), b[3].detach().cpu().numpy())
    assert G._REUSE_STATS["hits"] == hits              # (a's entry was ruled out: c ran its own forward)


def test_two_host_threads_drive_their_own_gaussian_sets_concurrently():
    """include/gdr.h promises thread safety for distinct workspaces, and the compiled boundary (csrc/boundary.cpp) keeps its
    group registry, reuse history and pace counters behind mutexes / per thread: two host threads, each running the reference's
    per-view loop + one backward on its OWN leaves and its own stream, several steps, must each get exactly what the same loop
    gives alone — images bit for bit, gradients within the per-element bar (the K7 atomics' order is the only freedom)."""
    import threading
    dev, base, sets, tg, n = _setup(V=4, n=12_000, B=2, seed=17)
    want = {}
    for i in (0, 1):
        leaves = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        imgs, losses, _ = _reference_loop(leaves, sets, tg, n, dev, i=i)
        sum(losses).backward()
        torch.cuda.synchronize()
        want[i] = ([x.detach().cpu().numpy() for x in imgs], {k: v.grad.cpu().numpy() for k, v in leaves.items()})
    got, errors = {}, []

    def worker(i):
        try:
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                for _ in range(6):
                    leaves = {k: v.clone().requires_grad_(True) for k, v in base.items()}
                    imgs, losses, _ = _reference_loop(leaves, sets, tg, n, dev, i=i)
                    sum(losses).backward()
                stream.synchronize()
                got[i] = ([x.detach().cpu().numpy() for x in imgs], {k: v.grad.cpu().numpy() for k, v in leaves.items()})
        except Exception as exc:      # noqa: BLE001
            errors.append((i, repr(exc)))
    torch.cuda.synchronize()
    threads = [threading.Thread(target=worker, args=(i,)) for i in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in (0, 1):
        for a, b in zip(got[i][0], want[i][0]):
            np.testing.assert_array_equal(a, b)
        for k in want[i][1]:
            out, worst, maxn = U.elem_stats(got[i][1][k], want[i][1][k])
            assert out < U.MAX_OUTSIDE and maxn < 1e-4, (i, k, out, worst, maxn)
            assert np.abs(got[i][1][k][1 - i]).max() == 0          # the other sample was never rendered by this thread
