"""Python boundary of the MI355X 2D-Gaussian (surfel) rasterizer — same surface as the `diff_surfel_rasterization`
package the reference's 2DGS adaptor imports (/root/reference/lightning/renderer_2dgs.py:7-10):

    GaussianRasterizationSettings   the same 12-field NamedTuple as the 3DGS path (renderer_2dgs.py:111-124)
    GaussianRasterizer(nn.Module)   forward(means3D, means2D, opacities, shs=, colors_precomp=, scales=, rotations=,
                                    cov3D_precomp=) -> (color (3,H,W), radii (N) int32, allmap (7,H,W))
                                    (renderer_2dgs.py:224-234); scales are (N,2); `cov3D_precomp` carries a
                                    precomputed (N,9) / (N,3,3) splat-to-pixel matrix
    rasterize_gaussians(...)        functional form

All arithmetic is in libgdr_hip.so (include/gsr.h) through ctypes; there is no CPU path: non-HIP tensors raise.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from . import _lib as L
from .rasterizer import (GaussianRasterizationSettings, _f32, _ptr, _require_hip, _settings_struct, _State,
                         _stream)
from . import _debug as _K
from . import rasterizer as _R
from . import viewgroup


class _SurfelState(_State):
    """Typed views of the surfel workspaces (rec is (N,24); n_contrib (2,H,W); final_T (3,H,W))."""

    __slots__ = ()

    def tensors(self) -> dict:
        N, D, H, W = max(self.N, 1), self.D, self.H, self.W
        g, b, im = self.geom, self.bin, self.img
        gb, bb, ib = self.geom_buf, self.bin_buf, self.img_buf
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        rec = self._view(gb, g.rec, torch.float32, L.GSR_REC_FLOATS * N).view(N, L.GSR_REC_FLOATS)
        out = dict(
            depths=self._view(gb, g.depths, torch.float32, N), rec=rec,
            rect=self._view(gb, g.rect, torch.int32, 4 * N).view(N, 4),
            tiles_touched=self._view(gb, g.tiles_touched, torch.int32, N),
            clamped=self._view(gb, g.clamped, torch.uint8, N),
            ranges=self._view(ib, im.ranges, torch.int32, 2 * tiles).view(tiles, 2),
            n_contrib=self._view(ib, im.n_contrib, torch.int32, 2 * H * W).view(2, H, W),
            final_T=self._view(ib, im.final_T, torch.float32, 3 * H * W).view(3, H, W),
            num_rendered=D,
            transMats=torch.cat([rec[:, 0:3], rec[:, 4:7], rec[:, 8:11]], 1),
            xy=torch.stack([rec[:, 3], rec[:, 7]], 1),
            normal_opacity=torch.cat([rec[:, 12:15], rec[:, 11:12]], 1),
            rgb=rec[:, 15:18], box=rec[:, 18:22], seg_len=int(b.seg_len),
            seg_count=self._view(bb, b.seg_count, torch.int32, 2))
        s = b.sorted
        if D > 0:
            out["point_list"] = self._view(bb, b.values[s], torch.int32, D)
            out["keys_sorted"] = _R.sorted_keys(out["ranges"], out["point_list"], out["depths"])
        else:
            out["keys_sorted"] = torch.empty(0, dtype=torch.int64, device=gb.device)
            out["point_list"] = torch.empty(0, dtype=torch.int32, device=gb.device)
        return out


def _inputs_struct(N, M, means3D, opacities, sh, colors_precomp, scales, rotations, transmat, flags=0) -> L.GsrInputs:
    return L.GsrInputs(N, M, _ptr(means3D), _ptr(opacities), _ptr(sh), _ptr(colors_precomp), _ptr(scales),
                       _ptr(rotations), _ptr(transmat), int(flags), 0)


def forward_raw(means3D, sh, colors_precomp, opacities, scales, rotations, transMat_precomp, raster_settings, flags=0,
                same_as=None):
    """Un-differentiated forward.  Returns (color, radii, allmap, state, keep)."""
    lib = L.load()
    _require_hip(means3D, "means3D")
    dev = means3D.device
    means3D, opacities, sh = _f32(means3D, dev), _f32(opacities, dev), _f32(sh, dev)
    colors_precomp, scales, rotations = _f32(colors_precomp, dev), _f32(scales, dev), _f32(rotations, dev)
    transMat_precomp = _f32(transMat_precomp, dev)
    N = int(means3D.shape[0])
    M = int(sh.shape[1]) if sh.numel() else 0
    H, W = int(raster_settings.image_height), int(raster_settings.image_width)
    if opacities.numel() != N:
        raise RuntimeError("opacities must have N elements")
    if scales.numel() and scales.shape[-1] != 2:
        raise RuntimeError("diff_surfel_rasterization: scales must be (N,2)")
    if transMat_precomp.numel() and transMat_precomp.numel() != 9 * N:
        raise RuntimeError("diff_surfel_rasterization: precomputed transform must be (N,9) or (N,3,3)")
    keep = [means3D, opacities, sh, colors_precomp, scales, rotations, transMat_precomp, int(flags)]
    with torch.cuda.device(dev):
        s = _settings_struct(raster_settings, dev, keep)
        inp = _inputs_struct(N, M, means3D, opacities, sh, colors_precomp, scales, rotations, transMat_precomp, flags)
        f32 = dict(dtype=torch.float32, device=dev)
        color = torch.empty(3, H, W, **f32)
        allmap = torch.empty(7, H, W, **f32)
        radii = torch.empty(N, dtype=torch.int32, device=dev)
        out = L.GsrOutputs(color.data_ptr(), allmap.data_ptr(), _ptr(radii))
        # one native call (include/gsr.h gsr_forward_view): carving, K1s, binning, K6s, the count read-back and the
        # per-shape history are the library's (rasterizer.forward_view_native)
        ws, vs = _R.forward_view_native(lib.gsr_forward_view, C.byref(s), C.byref(inp), N, H, W, True, C.byref(out), same_as,
                                        dev, _stream())
        st = _SurfelState()
        st.N, st.M, st.H, st.W, st.D = N, M, H, W, int(vs.D)
        st.geom_buf = st.bin_buf = st.img_buf = ws
        st.view, st.geom, st.bin, st.img = vs, vs.geom, vs.bin, vs.img
        keep.append(s)      # (reused by the backward of this call)
        if same_as and vs.differ:
            raise _R.GroupMismatch("diff_surfel_rasterization: equal autograd provenance but different values")
    return color, radii, allmap, st, keep


def backward_raw(st, keep, raster_settings, radii, grad_color, grad_allmap):
    """Returns dict of gradients (all fp32, on the inputs' device)."""
    lib = L.load()
    means3D, opacities, sh, colors_precomp, scales, rotations, transmat, flags = keep[:8]
    dev = means3D.device
    N, M = st.N, st.M
    f32 = dict(dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        keep2: list = []
        s = keep[-1] if isinstance(keep[-1], L.GdrSettings) else _settings_struct(raster_settings, dev, keep2)
        inp = _inputs_struct(N, M, means3D, opacities, sh, colors_precomp, scales, rotations, transmat, flags)
        gc = _f32(grad_color, dev)
        ga = None if grad_allmap is None else _f32(grad_allmap, dev)
        use_sh, use_tm = sh.numel() > 0, transmat.numel() > 0
        g = dict(
            means3D=torch.empty(N, 3, **f32), means2D=torch.empty(N, 4, **f32),
            shs=torch.empty(N, M, 3, **f32) if use_sh else None,
            colors_precomp=None if use_sh else torch.empty(N, 3, **f32),
            opacities=torch.empty(N, 1, **f32),
            scales=None if use_tm else torch.empty(N, 2, **f32),
            rotations=None if use_tm else torch.empty(N, 4, **f32),
            transMat_precomp=torch.empty(N, 9, **f32) if use_tm else None)
        scratch = torch.empty(max(N, 1) * L.GSR_GRAD_FLOATS, **f32)
        gin = L.GsrGradInputs(gc.data_ptr(), _ptr(ga))
        gout = L.GsrGradOutputs(_ptr(g["means3D"]), _ptr(g["means2D"]), _ptr(g["shs"]), _ptr(g["colors_precomp"]),
                                _ptr(g["opacities"]), _ptr(g["scales"]), _ptr(g["rotations"]),
                                _ptr(g["transMat_precomp"]), scratch.data_ptr(), 0, 0)
        L.check(lib.gsr_backward(C.byref(s), C.byref(inp), C.byref(st.geom), C.byref(st.bin), C.byref(st.img), st.D,
                                 _ptr(radii), C.byref(gin), C.byref(gout), _stream()), "gsr_backward")
    return g


class _RasterizeSurfels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, transMat_precomp,
                raster_settings, flags=0):
        color, radii, allmap, st, keep = forward_raw(means3D, sh, colors_precomp, opacities, scales, rotations,
                                                     transMat_precomp, raster_settings, flags)
        ctx.raster_settings, ctx.state, ctx.radii = raster_settings, st, radii
        ctx.pace = viewgroup.pace()
        _R._save_inputs(ctx, keep)
        ctx.means2D_shape = tuple(means2D.shape)
        ctx.tm_shape = tuple(transMat_precomp.shape)
        ctx.in_dtypes = tuple(t.dtype for t in (means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                                transMat_precomp))
        ctx.mark_non_differentiable(radii)
        # an output the loss does not use arrives as None instead of a zero tensor: a missing allmap gradient (the reference's
        # fine-stage renders, lightning/loss.py:35-50) selects the image-only K7s
        ctx.set_materialize_grads(False)
        return color, radii, allmap

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_allmap):
        viewgroup.note_backward(ctx.pace)
        if grad_color is None:
            grad_color = torch.zeros(3, ctx.state.H, ctx.state.W, dtype=torch.float32, device=ctx.radii.device)
        g = backward_raw(ctx.state, _R._saved_inputs(ctx), ctx.raster_settings, ctx.radii, grad_color, grad_allmap)
        gm2 = g["means2D"]
        cols = ctx.means2D_shape[1] if len(ctx.means2D_shape) == 2 else 4
        if cols == 3:  # upstream's (N,3) carrier: xy signal, z = 0
            gm2 = torch.cat([gm2[:, :2], torch.zeros_like(gm2[:, :1])], dim=1)
        elif cols != 4:
            gm2 = gm2[:, :cols].contiguous()
        gtm = g["transMat_precomp"]
        if gtm is not None:
            gtm = gtm.reshape(ctx.tm_shape)
        grads = [g["means3D"], gm2, g["shs"], g["colors_precomp"], g["opacities"], g["scales"], g["rotations"], gtm]
        grads = [None if t is None else (t if t.dtype == dt else t.to(dt)) for t, dt in zip(grads, ctx.in_dtypes)]
        return (*grads, None, None)


# --------------------------------------------------------------------------------------------
# Multi-view node (SURVEY §8f-1 for the surfel path): V views of ONE surfel set in one autograd node — K1s for every
# view without a host sync, ONE read-back of the V duplicate counts, binning + K6s per view; backward runs K7s + K9s
# per view with K9s ACCUMULATING into one set of gradient buffers (no autograd accumulation passes).
# --------------------------------------------------------------------------------------------
def _surfel_forward_views_impl(ctx, means3D, means2D, sh, opacities, scales, rotations, settings_list, flags, loss_spec=None):
        """Forward of the multi-view surfel nodes.  loss_spec = (rays, views, targets, weights, losses): the per-view
        image loss (gsr_view_loss_forward) is enqueued right behind the view's K6s on the same stream."""
        lib = L.load()
        _require_hip(means3D, "means3D")
        dev = means3D.device
        in_dtypes = tuple(t.dtype for t in (means3D, means2D, sh, opacities, scales, rotations))
        means3D, sh = _f32(means3D, dev), _f32(sh, dev)
        opacities, scales, rotations = _f32(opacities, dev), _f32(scales, dev), _f32(rotations, dev)
        N, M, V = int(means3D.shape[0]), int(sh.shape[1]), len(settings_list)
        H, W = int(settings_list[0].image_height), int(settings_list[0].image_width)
        if scales.shape[-1] != 2:
            raise RuntimeError("diff_surfel_rasterization: scales must be (N,2)")
        e = torch.empty(0, dtype=torch.float32, device=dev)
        keep = [means3D, opacities, sh, e, scales, rotations, e, int(flags)]
        f32, u8 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.uint8, device=dev)
        sizes = [(int(rs.image_height), int(rs.image_width)) for rs in settings_list]   # views may differ in size
        colors = [torch.empty(3, h, w, **f32) for h, w in sizes]
        allmaps = [torch.empty(7, h, w, **f32) for h, w in sizes]
        radii = torch.empty(V, N, dtype=torch.int32, device=dev)

        def view_loss(v, sv):  # loss of view v from (color, allmap), on the stream its K6s runs on
            if loss_spec is None:
                return
            rays, views, targets, wts, losses = loss_spec
            L.check(lib.gsr_view_loss_forward(colors[v].data_ptr(), allmaps[v].data_ptr(), rays[v].data_ptr(),
                                              views[v].data_ptr(), targets[v].data_ptr(), H, W, *wts,
                                              losses[v:v + 1].data_ptr(), sv), "gsr_view_loss_forward")

        states, structs = [], []
        # duplicate counters of the V views in one array; their read-back does not stall the call once the shape has a
        # history (rasterizer.DEFER_D: device-sized binning calls, capacity check after everything is enqueued)
        counters = torch.empty(V, dtype=torch.int32, device=dev)
        key = ("surfel",) + _R.shape_key(N, H, W, V)
        with torch.cuda.device(dev):
            stream = _stream()
            main = torch.cuda.current_stream()
            # the gradient records of a same-size call: allocated now, cleared at the end of this forward on the K7 streams
            # (rasterizer._early_records_begin: the event those streams wait for must precede the join with the side streams)
            pending_recs = (_R._early_records_begin(ctx, dev, V, N, H, W, L.GSR_GRAD_FLOATS)
                            if all(int(rs.image_height) == H and int(rs.image_width) == W for rs in settings_list) else None)
            inp = _inputs_struct(N, M, means3D, opacities, sh, e, scales, rotations, e, flags)
            for v, rs in enumerate(settings_list):
                s = _settings_struct(rs, dev, keep)
                st = _SurfelState()
                st.N, st.M, st.H, st.W = N, M, int(rs.image_height), int(rs.image_width)
                st.geom_buf = torch.empty(lib.gsr_geom_bytes(N), **u8)
                st.img_buf = torch.empty(lib.gsr_image_bytes(st.H, st.W), **u8)
                st.bin_buf = None
                st.geom, st.bin, st.img = L.GdrGeom(), L.GdrBinning(), L.GdrImage()
                L.check(lib.gsr_geom_carve(st.geom_buf.data_ptr(), N, C.byref(st.geom)), "gsr_geom_carve")
                L.check(lib.gsr_image_carve(st.img_buf.data_ptr(), st.H, st.W, C.byref(st.img)), "gsr_image_carve")
                st.geom.num_rendered = counters.data_ptr() + 4 * v
                st.counters = counters
                states.append(st)
                structs.append(s)
            same = all(int(rs.image_height) == H and int(rs.image_width) == W for rs in settings_list)
            if same:   # K1s for all views in groups of <= GDR_MAX_VIEWS launches (inputs read once per group)
                for lo in range(0, V, L.GDR_MAX_VIEWS):
                    n = min(L.GDR_MAX_VIEWS, V - lo)
                    s_arr = (L.GdrSettings * n)(*structs[lo:lo + n])
                    g_arr = (L.GdrGeom * n)(*[states[lo + k].geom for k in range(n)])
                    r_arr = (C.c_void_p * n)(*[radii[lo + k].data_ptr() if N else None for k in range(n)])
                    L.check(lib.gsr_preprocess_forward_views(n, s_arr, C.byref(inp), g_arr, r_arr, stream),
                            "gsr_preprocess_forward_views")
            else:
                for v, st in enumerate(states):
                    L.check(lib.gsr_preprocess_forward(C.byref(structs[v]), C.byref(inp), C.byref(st.geom), _ptr(radii[v]),
                                                       None, stream), "gsr_preprocess_forward")
            # every view's chain (binning -> K6s -> fused loss kernel) on one of the forward streams, the caller's
            # included (rasterizer._forward_views_impl)
            nfs = max(1, min(_K.FWD_STREAMS, V)) if _K.RENDER_SIDE and V > 1 and _R.side_count(H, W) > 0 else 1
            fstreams = [main] + _R._view_streams(dev, nfs - 1)
            if nfs > 1:
                ready = torch.cuda.Event()
                ready.record(main)
                for fs in fstreams[1:]:
                    fs.wait_event(ready)
            readback = _R._CountReadback(counters, fstreams[-1])
            cap = _R._d_capacity(key, N) if N > 0 else None
            tiles_of = lambda st: ((st.W + 15) // 16) * ((st.H + 15) // 16)
            stats, hints = _R._launch_stats(key, V)
            srow = (lambda v: None) if stats is None else (lambda v: stats[v])
            if cap is None:
                d_host = readback.wait()
                for v, st in enumerate(states):
                    _R._carve_binning(lib, st, d_host[v], tiles_of(st), stats=srow(v), hints=hints)
            else:
                for v, st in enumerate(states):
                    _R._carve_binning(lib, st, cap, tiles_of(st), d_dev=st.geom.num_rendered, stats=srow(v), hints=hints)

            def chain(v, fs):
                st, sv = states[v], C.c_void_p(fs.cuda_stream)
                L.check(lib.gdr_binning_forward(C.byref(structs[v]), N, C.byref(st.geom), C.byref(st.bin), C.byref(st.img),
                                                st.D, _ptr(radii[v]), sv), "gdr_binning_forward")
                out = L.GsrOutputs(colors[v].data_ptr(), allmaps[v].data_ptr(), _ptr(radii[v]))
                L.check(lib.gsr_composite_forward(C.byref(structs[v]), C.byref(st.geom), C.byref(st.bin), C.byref(st.img),
                                                  C.byref(out), sv), "gsr_composite_forward")
                view_loss(v, sv)

            for v in range(V):
                chain(v, fstreams[v % nfs])
            for fs in fstreams[1:]:
                done = torch.cuda.Event()
                done.record(fs)
                main.wait_event(done)
            if cap is not None:
                d_host = readback.wait()
                for v in [v for v in range(V) if d_host[v] > cap]:   # the capacity guess was too small: repeat the view
                    _R._carve_binning(lib, states[v], d_host[v], tiles_of(states[v]), stats=srow(v), hints=hints)
                    if loss_spec is not None:
                        loss_spec[-1][v:v + 1].zero_()
                    chain(v, main)
                for v, st in enumerate(states):
                    st.D = d_host[v]
            _R._d_record(key, d_host, N)
        ctx.states, ctx.settings_list, ctx.radii, ctx.flags = states, settings_list, radii, int(flags)
        _R._save_inputs(ctx, keep)
        ctx.means2D_shape, ctx.in_dtypes, ctx.V = tuple(means2D.shape), in_dtypes, V
        ctx.mark_non_differentiable(radii)
        ctx.recs = None
        if same:   # (the records of the fused backward: cleared now, under the tail of the forward)
            _R._early_records_clear(ctx, pending_recs)
        return radii, colors, allmaps


class _RenderSurfelViews(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, opacities, scales, rotations, settings_list, flags):
        radii, colors, allmaps = _surfel_forward_views_impl(ctx, means3D, means2D, sh, opacities, scales, rotations,
                                                            settings_list, flags)
        return (radii, *colors, *allmaps)

    @staticmethod
    def backward(ctx, g_radii, *g):
        lib = L.load()
        V = ctx.V
        means3D, opacities, sh, _, scales, rotations, _, flags = _R._saved_inputs(ctx)[:8]
        dev = means3D.device
        N, M = int(means3D.shape[0]), int(sh.shape[1])
        f32 = dict(dtype=torch.float32, device=dev)
        out = dict(means3D=torch.empty(N, 3, **f32), means2D=torch.empty(N, 4, **f32), shs=torch.empty(N, M, 3, **f32),
                   opacities=torch.empty(N, 1, **f32), scales=torch.empty(N, 2, **f32), rotations=torch.empty(N, 4, **f32))
        e = torch.empty(0, dtype=torch.float32, device=dev)
        first = True
        nb = (int(ctx.settings_list[0].sh_degree) + 1) ** 2
        fused = (all(x is not None for x in g[:V]) and not ((3 * nb) % 4 == 0 and M != nb)
                 and all(st.H == ctx.states[0].H and st.W == ctx.states[0].W for st in ctx.states))
        if fused:   # K7s per view into its own record, then ONE K9s launch per <= 8 views
            with torch.cuda.device(dev):
                stream = _stream()
                keep2: list = []
                inp = _inputs_struct(N, M, means3D, opacities, sh, e, scales, rotations, e, flags)
                sets, gins = [], []
                for v in range(V):  # torch-side preparation on the caller's stream first
                    sets.append(_settings_struct(ctx.settings_list[v], dev, keep2))
                    gc = _f32(g[v], dev)
                    ga = None if g[V + v] is None else _f32(g[V + v], dev)
                    keep2 += [gc, ga]
                    gins.append(L.GsrGradInputs(gc.data_ptr(), _ptr(ga)))
                # K7s of the views: ONE launch per <= 8 views on the caller's stream (round 4, as the 3DGS node: K.K7_VIEWS), or
                # one launch per view on side streams (mode 0); K9s per group behind its views' K7s
                mode = _R.k7_views_mode(ctx.states[0].H, ctx.states[0].W, N) if V > 1 else 0
                if mode:
                    _R._join_record_clears(ctx)
                sides = _R._SideViews(dev, 1 if mode else V, ctx.states[0].H, ctx.states[0].W)
                for lo in range(0, V, L.GDR_MAX_VIEWS):
                    n = min(L.GDR_MAX_VIEWS, V - lo)
                    recs, cleared = _R._take_records(ctx, lo, n, N, L.GSR_GRAD_FLOATS, dev)
                    s_arr = (L.GdrSettings * n)(*sets[lo:lo + n])
                    g_arr, b_arr, i_arr = (L.GdrGeom * n)(), (L.GdrBinning * n)(), (L.GdrImage * n)()
                    for k in range(n):
                        st = ctx.states[lo + k]
                        st.bin.grad_rec_cleared = cleared
                        g_arr[k], b_arr[k], i_arr[k] = st.geom, st.bin, st.img
                    if mode:
                        gin_arr = (L.GsrGradInputs * n)(*gins[lo:lo + n])
                        rec_ptrs = (C.c_void_p * n)(*[recs[k].data_ptr() for k in range(n)])
                        L.check(lib.gsr_render_backward_views(n, s_arr, N, g_arr, b_arr, i_arr, gin_arr, rec_ptrs, int(mode == 1),
                                                              stream), "gsr_render_backward_views")
                    for k in range(0 if mode else n):
                        st = ctx.states[lo + k]
                        L.check(lib.gsr_render_backward(C.byref(s_arr[k]), N, C.byref(g_arr[k]), C.byref(st.bin),
                                                        C.byref(st.img), C.byref(gins[lo + k]), recs[k].data_ptr(),
                                                        sides.stream(lo + k)), "gsr_render_backward")
                    r_arr = (C.c_void_p * n)(*[ctx.radii[lo + k].data_ptr() if N else None for k in range(n)])
                    rec_arr = (C.c_void_p * n)(*[recs[k].data_ptr() for k in range(n)])
                    gout = L.GsrGradOutputs(_ptr(out["means3D"]), _ptr(out["means2D"]), _ptr(out["shs"]), None,
                                            _ptr(out["opacities"]), _ptr(out["scales"]), _ptr(out["rotations"]), None, None,
                                            1 if lo > 0 else 0, 0)
                    L.check(lib.gsr_preprocess_backward_views(n, s_arr, C.byref(inp), g_arr, r_arr, rec_arr, C.byref(gout),
                                                              sides.k9_stream(lo, n)), "gsr_preprocess_backward_views")
                    keep2.append(recs)
                sides.join()
            first = False
        scratch = torch.empty(max(N, 1) * L.GSR_GRAD_FLOATS, **f32) if not fused else None
        with torch.cuda.device(dev):
            stream = _stream()
            inp = _inputs_struct(N, M, means3D, opacities, sh, e, scales, rotations, e, flags)
            for v in range(V if not fused else 0):
                gc, ga = g[v], g[V + v]
                if gc is None and ga is None:
                    continue
                st = ctx.states[v]
                gc = torch.zeros(3, st.H, st.W, **f32) if gc is None else _f32(gc, dev)
                ga = None if ga is None else _f32(ga, dev)
                keep2: list = []
                s = _settings_struct(ctx.settings_list[v], dev, keep2)
                gin = L.GsrGradInputs(gc.data_ptr(), _ptr(ga))
                gout = L.GsrGradOutputs(_ptr(out["means3D"]), _ptr(out["means2D"]), _ptr(out["shs"]), None,
                                        _ptr(out["opacities"]), _ptr(out["scales"]), _ptr(out["rotations"]), None,
                                        scratch.data_ptr(), 0 if first else 1, 0)
                L.check(lib.gsr_backward(C.byref(s), C.byref(inp), C.byref(st.geom), C.byref(st.bin), C.byref(st.img), st.D,
                                         _ptr(ctx.radii[v]), C.byref(gin), C.byref(gout), stream), "gsr_backward")
                first = False
        if first:
            for t in out.values():
                t.zero_()
        gm2 = out["means2D"]
        cols = ctx.means2D_shape[1] if len(ctx.means2D_shape) == 2 else 4
        if cols == 3:
            gm2 = torch.cat([gm2[:, :2], torch.zeros_like(gm2[:, :1])], dim=1)
        elif cols != 4:
            gm2 = gm2[:, :cols].contiguous()
        grads = [out["means3D"], gm2, out["shs"], out["opacities"], out["scales"], out["rotations"]]
        grads = [t if t.dtype == dt else t.to(dt) for t, dt in zip(grads, ctx.in_dtypes)]
        return (*grads, None, None)


class _RenderSurfelViewsLoss(torch.autograd.Function):
    """V views of one surfel set AND their image losses in one node (SURVEY §8f-4 for the 2DGS path): the fused loss
    kernels of a view (gsr_view_loss_forward / _backward: the adaptor's maps + clamp + MSE + distortion + normal
    consistency + depth / alpha means) run on the view's side stream right behind its K6s / in front of its K7s, so
    they overlap the other views' render kernels instead of running alone between the forward and the backward; no
    dL/dimage tensors cross autograd.  Returns (losses (V,), radii (V,N))."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, opacities, scales, rotations, settings_list, flags, rays, views, targets, wts):
        dev = means3D.device
        V = len(settings_list)
        H, W = int(settings_list[0].image_height), int(settings_list[0].image_width)
        if any(int(rs.image_height) != H or int(rs.image_width) != W for rs in settings_list):
            raise RuntimeError("render_views_loss: all views must share one image size")
        rays = [r.to(device=dev, dtype=torch.float32).contiguous() for r in rays]
        views = [m.to(device=dev, dtype=torch.float32).contiguous() for m in views]
        targets = [t.to(device=dev, dtype=torch.float32).contiguous() for t in targets]
        losses = torch.zeros(V, dtype=torch.float32, device=dev)
        wts = tuple(float(x) for x in wts)
        radii, colors, allmaps = _surfel_forward_views_impl(ctx, means3D, means2D, sh, opacities, scales, rotations,
                                                            settings_list, flags, loss_spec=(rays, views, targets, wts, losses))
        ctx.loss_in = (colors, allmaps, rays, views, targets, wts)
        return losses, radii

    @staticmethod
    def backward(ctx, g_losses, g_radii):
        lib = L.load()
        V = ctx.V
        colors, allmaps, rays, views, targets, wts = ctx.loss_in
        means3D, opacities, sh, _, scales, rotations, _, flags = _R._saved_inputs(ctx)[:8]
        dev = means3D.device
        N, M = int(means3D.shape[0]), int(sh.shape[1])
        H, W = ctx.states[0].H, ctx.states[0].W
        f32 = dict(dtype=torch.float32, device=dev)
        out = dict(means3D=torch.empty(N, 3, **f32), means2D=torch.empty(N, 4, **f32), shs=torch.empty(N, M, 3, **f32),
                   opacities=torch.empty(N, 1, **f32), scales=torch.empty(N, 2, **f32), rotations=torch.empty(N, 4, **f32))
        e = torch.empty(0, dtype=torch.float32, device=dev)
        go = g_losses.to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            stream = _stream()
            keep2: list = []
            inp = _inputs_struct(N, M, means3D, opacities, sh, e, scales, rotations, e, flags)
            sets = [_settings_struct(rs, dev, keep2) for rs in ctx.settings_list]
            # the loss-backward kernels of the views run on side streams (they overlap each other); K7s of the views then
            # in ONE launch per <= 8 views on the caller's stream behind them (round 4; mode 0: per view on the side streams)
            mode = _R.k7_views_mode(H, W, N) if V > 1 else 0
            sides = _R._SideViews(dev, V, H, W)  # after every torch-side preparation
            for lo, n in sides.groups():
                recs, cleared = _R._take_records(ctx, lo, n, N, L.GSR_GRAD_FLOATS, dev)
                dcs = [torch.empty(3, H, W, **f32) for _ in range(n)]
                das = [torch.empty(7, H, W, **f32) for _ in range(n)]
                scr = [torch.empty(9, H, W, **f32) for _ in range(n)]
                s_arr = (L.GdrSettings * n)(*sets[lo:lo + n])
                g_arr, b_arr, i_arr = (L.GdrGeom * n)(), (L.GdrBinning * n)(), (L.GdrImage * n)()
                gin_arr = (L.GsrGradInputs * n)()
                for k in range(n):
                    v = lo + k
                    st = ctx.states[v]
                    st.bin.grad_rec_cleared = cleared
                    g_arr[k], b_arr[k], i_arr[k] = st.geom, st.bin, st.img
                    sv = sides.stream(v)
                    L.check(lib.gsr_view_loss_backward(colors[v].data_ptr(), allmaps[v].data_ptr(), rays[v].data_ptr(),
                                                       views[v].data_ptr(), targets[v].data_ptr(), H, W, *wts,
                                                       go[v:v + 1].data_ptr(), scr[k].data_ptr(), dcs[k].data_ptr(),
                                                       das[k].data_ptr(), sv), "gsr_view_loss_backward")
                    gin_arr[k] = L.GsrGradInputs(dcs[k].data_ptr(), das[k].data_ptr())
                    if not mode:
                        L.check(lib.gsr_render_backward(C.byref(s_arr[k]), N, C.byref(g_arr[k]), C.byref(st.bin),
                                                        C.byref(st.img), C.byref(gin_arr[k]), recs[k].data_ptr(), sv),
                                "gsr_render_backward")
                if mode:   # (k9_stream: the caller's stream, made to wait for the side streams — the early record clears
                    # queued there before the loss kernels included)
                    rec_ptrs = (C.c_void_p * n)(*[recs[k].data_ptr() for k in range(n)])
                    L.check(lib.gsr_render_backward_views(n, s_arr, N, g_arr, b_arr, i_arr, gin_arr, rec_ptrs, int(mode == 1),
                                                          sides.k9_stream(lo, n)), "gsr_render_backward_views")
                r_arr = (C.c_void_p * n)(*[ctx.radii[lo + k].data_ptr() if N else None for k in range(n)])
                rec_arr = (C.c_void_p * n)(*[recs[k].data_ptr() for k in range(n)])
                gout = L.GsrGradOutputs(_ptr(out["means3D"]), _ptr(out["means2D"]), _ptr(out["shs"]), None,
                                        _ptr(out["opacities"]), _ptr(out["scales"]), _ptr(out["rotations"]), None, None,
                                        1 if lo > 0 else 0, 0)
                L.check(lib.gsr_preprocess_backward_views(n, s_arr, C.byref(inp), g_arr, r_arr, rec_arr, C.byref(gout),
                                                          sides.k9_stream(lo, n)), "gsr_preprocess_backward_views")
                keep2 += [recs, dcs, das, scr]
            sides.join()
        gm2 = out["means2D"]
        cols = ctx.means2D_shape[1] if len(ctx.means2D_shape) == 2 else 4
        if cols == 3:
            gm2 = torch.cat([gm2[:, :2], torch.zeros_like(gm2[:, :1])], dim=1)
        elif cols != 4:
            gm2 = gm2[:, :cols].contiguous()
        grads = [out["means3D"], gm2, out["shs"], out["opacities"], out["scales"], out["rotations"]]
        grads = [t if t.dtype == dt else t.to(dt) for t, dt in zip(grads, ctx.in_dtypes)]
        return (*grads, None, None, None, None, None, None)


def render_surfel_views_loss_raw(means3D, means2D, sh, opacities, scales, rotations, settings_list, rays, viewmatrices,
                                 targets_chw, depth_ratio=0.0, w_dist=1000.0, w_normal=0.2, w_depth=0.1, w_alpha=0.1,
                                 flags=0):
    """Per-view losses (V,) of V views of one surfel set, the loss kernels folded into the node (see the class).
    rays: V tensors (H,W,6); viewmatrices: V world_view_transform; targets_chw: V tensors (3,H,W).  (losses, radii)."""
    nb = (int(settings_list[0].sh_degree) + 1) ** 2
    if (3 * nb) % 4 == 0 and int(sh.shape[1]) != nb:
        raise RuntimeError("render_views_loss: shs must hold exactly (sh_degree+1)^2 coefficients")
    return _RenderSurfelViewsLoss.apply(means3D, means2D, sh, opacities, scales, rotations, list(settings_list), int(flags),
                                        list(rays), list(viewmatrices), list(targets_chw),
                                        (depth_ratio, w_dist, w_normal, w_depth, w_alpha))


def render_surfel_views_raw(means3D, means2D, sh, opacities, scales, rotations, settings_list, flags=0):
    """All views of one surfel set in one node.  opacities / scales (N,2) / rotations are RAW when the matching
    GDR_IN_RAW_* flag is set.  Returns (colors [V x (3,H,W)], radii (V,N), allmaps [V x (7,H,W)])."""
    V = len(settings_list)
    res = _RenderSurfelViews.apply(means3D, means2D, sh, opacities, scales, rotations, list(settings_list), int(flags))
    return list(res[1:1 + V]), res[0], list(res[1 + V:1 + 2 * V])


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    """The reference boundary (/root/reference/lightning/renderer_2dgs.py:224-234).  Calls that are provably handed the same
    surfels as earlier ones join a render group — one K9s for all of them (viewgroup.py, round 4); everything else is one
    independent autograd node per call."""
    if not torch.is_grad_enabled():     # evaluation: no autograd node at all
        return forward_raw(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings)[:3]
    viewgroup.note_forward()
    if (viewgroup.eligible(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
            and scales.shape[-1] == 2 and viewgroup.PATH_SURFEL.supports(sh, raster_settings)):
        out = viewgroup.grouped_call(viewgroup.PATH_SURFEL, means3D, means2D, sh, opacities, scales, rotations,
                                     raster_settings)
        if out is not None:
            return out
    return _RasterizeSurfels.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                   cov3Ds_precomp, raster_settings)


class GaussianRasterizer(_R._LazyModule):      # (an nn.Module set up on first use: rasterizer._LazyModule)

    def markVisible(self, positions):
        return _R.GaussianRasterizer(self.raster_settings).markVisible(positions)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        e = _R.empty_f32(means3D.device)
        return rasterize_gaussians(
            means3D, means2D, e if shs is None else shs, e if colors_precomp is None else colors_precomp, opacities,
            e if scales is None else scales, e if rotations is None else rotations,
            e if cov3D_precomp is None else cov3D_precomp, self.raster_settings)
