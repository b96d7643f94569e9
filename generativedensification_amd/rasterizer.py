"""Python boundary of the MI355X rasterizer — same surface as the reference's
`diff_gaussian_rasterization` package (imported at /root/reference/lightning/renderer.py:10-13
and .../point_decoder/layers/gaussian_renderer.py:14):

    GaussianRasterizationSettings   12-field NamedTuple, built by keyword (renderer.py:111-124)
    GaussianRasterizer(nn.Module)   forward(means3D, means2D, opacities, shs=, colors_precomp=,
                                    scales=, rotations=, cov3D_precomp=)
                                    -> (color(3,H,W), radii(N) int32, depth(1,H,W), alpha(1,H,W))
                                    (renderer.py:250-259)
    rasterize_gaussians(...)        functional form (BASELINE.json north_star names it)
    GaussianRasterizer.markVisible  upstream API, unused by the reference

Host code is Python on PyTorch-ROCm (tensors, streams); all arithmetic is in
libgdr_hip.so, reached through the C ABI in include/gdr.h via ctypes.  There is no CPU
path: non-HIP tensors raise.
"""
from __future__ import annotations

import ctypes as C
import math
import os as _os
import threading
from typing import NamedTuple, Optional

import torch
from torch import nn

from . import _debug as K
from . import _lib as L
from . import viewgroup      # at module import (its torch probe must not wait for a first call under no_grad / in a backward)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None or t.numel() == 0 else C.c_void_p(t.data_ptr())


_EMPTY: dict = {}


def empty_f32(dev) -> torch.Tensor:
    """The (0,) float32 tensor standing for an absent optional input, one per device (never written)."""
    t = _EMPTY.get(dev)
    if t is None:
        t = _EMPTY[dev] = torch.empty(0, dtype=torch.float32, device=dev)
    return t


def _f32(t: torch.Tensor, dev) -> torch.Tensor:
    if t.device != dev:
        t = t.to(dev)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _require_hip(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"diff_gaussian_rasterization (MI355X build): `{name}` is on {t.device}; this "
            "rasterizer runs only on ROCm/HIP device tensors and has no CPU fallback.")


class _State:
    """Typed views of the three workspaces of one forward call (kept for backward and
    exposed to the parity tests)."""

    __slots__ = ("N", "M", "H", "W", "D", "geom_buf", "bin_buf", "img_buf", "geom", "bin", "img", "counters", "view")

    def _view(self, buf, ptr, dtype, count):
        off = ptr - buf.data_ptr()
        nbytes = count * torch.empty(0, dtype=dtype).element_size()
        return buf[off:off + nbytes].view(dtype)

    def tensors(self) -> dict:
        N, D, H, W = max(self.N, 1), self.D, self.H, self.W
        g, b, im = self.geom, self.bin, self.img
        gb, bb, ib = self.geom_buf, self.bin_buf, self.img_buf
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        s = b.sorted
        out = dict(
            depths=self._view(gb, g.depths, torch.float32, N),
            rec=self._view(gb, g.rec, torch.float32, 16 * N).view(N, 16),
            rect=self._view(gb, g.rect, torch.int32, 4 * N).view(N, 4),
            tiles_touched=self._view(gb, g.tiles_touched, torch.int32, N),
            clamped=self._view(gb, g.clamped, torch.uint8, N),
            ranges=self._view(ib, im.ranges, torch.int32, 2 * tiles).view(tiles, 2),
            n_contrib=self._view(ib, im.n_contrib, torch.int32, H * W).view(H, W),
            final_T=self._view(ib, im.final_T, torch.float32, H * W).view(H, W),
            num_rendered=D,
        )
        if gb.data_ptr() <= g.cov3D < gb.data_ptr() + gb.numel():   # (views > 0 of a multi-view node share view 0's)
            out["cov3D"] = self._view(gb, g.cov3D, torch.float32, 6 * N).view(N, 6)
        out["seg_len"] = int(b.seg_len)
        out["seg_count"] = self._view(bb, b.seg_count, torch.int32, 3)  # rows of seg_extra, state slots (filled by K6)
        out["xy"], out["conic_opacity"], out["rgb"] = out["rec"][:, 0:2], out["rec"][:, 4:8], out["rec"][:, 8:12]
        if D > 0:
            out["point_list"] = self._view(bb, b.values[s], torch.int32, D)
            out["keys_sorted"] = sorted_keys(out["ranges"], out["point_list"], out["depths"])
        else:
            out["keys_sorted"] = torch.empty(0, dtype=torch.int64, device=gb.device)
            out["point_list"] = torch.empty(0, dtype=torch.int32, device=gb.device)
        return out


def sorted_keys(ranges, point_list, depths):
    """The reference's sorted key list (tile << 32 | float bits of the depth) of a binned view, rebuilt
    from what the binning stage keeps: the per-tile ranges, the sorted ids and the per-Gaussian depths.  The kernels sort
    (depth bits, id) per tile and never materialise the 64-bit keys; the parity tests compare this list with the oracle's."""
    D = int(point_list.numel())
    r = ranges.long()
    tile_of = torch.repeat_interleave(torch.arange(r.shape[0], device=r.device), (r[:, 1] - r[:, 0]).clamp_min(0))
    if tile_of.numel() != D:       # (a truncated device-sized call: the caller repeats the view)
        tile_of = torch.cat([tile_of, tile_of.new_zeros(max(0, D - tile_of.numel()))])[:D]
    bits = depths.view(torch.int32)[point_list.long()].long() & 0xFFFFFFFF
    return (tile_of << 32) | bits


def _settings_struct(rs: GaussianRasterizationSettings, dev, keep: list) -> L.GdrSettings:
    bg = _f32(rs.bg, dev)
    view = _f32(rs.viewmatrix, dev)
    proj = _f32(rs.projmatrix, dev)
    campos = _f32(rs.campos, dev)
    keep += [bg, view, proj, campos]
    return L.GdrSettings(int(rs.image_height), int(rs.image_width), float(rs.tanfovx), float(rs.tanfovy),
                         float(rs.scale_modifier), int(rs.sh_degree), int(bool(rs.prefiltered)),
                         int(bool(rs.debug)), bg.data_ptr(), view.data_ptr(), proj.data_ptr(),
                         campos.data_ptr())


# Parity-risk switch R1 (include/gdr.h GDR_IN_NO_DEPTH_TO_MEAN): True (default) = dL/d(depth image) also
# moves the Gaussian centres (depth_i = view-space z of the centre); False = it does not.  GDR_DEPTH_TO_MEAN=0 or
# rasterizer.DEPTH_TO_MEAN = False.
DEPTH_TO_MEAN = _os.environ.get("GDR_DEPTH_TO_MEAN", "1") != "0"


def _inputs_struct(N, M, means3D, opacities, sh, colors_precomp, scales, rotations, cov3Ds, flags=0) -> L.GdrInputs:
    flags = int(flags) | (0 if DEPTH_TO_MEAN else L.GDR_IN_NO_DEPTH_TO_MEAN)
    return L.GdrInputs(N, M, _ptr(means3D), _ptr(opacities), _ptr(sh), _ptr(colors_precomp), _ptr(scales),
                       _ptr(rotations), _ptr(cov3Ds), flags, 0)


def _seg_len_for(D, tiles=None, busy=None):
    """Host-side policy of the cut lists: the segment length a view's binning workspace is carved for
    (gdr_binning_carve_seg sizes the cut-list tables for it: 80 bytes per duplicate at 256, 40 at 512).  D: duplicates
    expected in the view; busy: tiles with >= 64 entries in the previous call of the shape (None: unknown).  512-entry
    segments pay on large images whose tiles are all busy with long lists (C4 +2.5 %, C5 +1.3 % over 256); scenes with
    few busy tiles (an object in front of a background) or short lists keep 256 (C2 shell +5 %, C5 shell +4 %, C4 shell
    +1.2 %, C2 +0.5 % over 512)."""
    if K.SEG_LEN is not None:
        return max(0, int(K.SEG_LEN)) // 256 * 256
    if tiles is not None and tiles >= 2000 and D >= 500 * tiles and (busy is None or 2 * busy >= tiles):
        return 512
    return L.GDR_DEFAULT_SEG_LEN


def side_count(H, W):
    """Side streams for the views of one node: 2 at >= 2000 tiles per view (800x800), 3 for smaller images whose
    single view cannot fill 256 CUs (512x512 = 1024 tiles)."""
    if K.BIN_STREAM is not None:
        return K.BIN_STREAM
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    return 2 if tiles >= 2000 else 3


def k7_views_mode(H, W, N):
    """How K7 of a multi-view node is launched (K.K7_VIEWS above)."""
    if K.K7_VIEWS is not None:
        return K.K7_VIEWS
    return 1


_side_streams: dict = {}


def _view_streams(dev, n):
    with _HIST_LOCK:
        pool = _side_streams.setdefault(dev.index, [])
        while len(pool) < n:
            pool.append(torch.cuda.Stream(device=dev))
        return pool[:n]


_raw_stream = torch._C._cuda_getCurrentRawStream   # (torch.cuda.current_stream() builds a Stream object: 9 us per call)


def _stream():
    return C.c_void_p(_raw_stream(torch.cuda.current_device()))


class _SideViews:
    """Streams of the backward of a multi-view node.  K7 of the views runs round-robin on side streams: the views are
    independent, one view's kernel leaves CUs idle whenever its tile lists are skewed (an object in front of an empty
    background: a few hundred busy tiles for 256 CUs) and at its tail, and another view's workgroups fill them.  K9 (one
    launch per <= GDR_MAX_VIEWS views, summing their records into the per-Gaussian gradients; every launch after the
    first accumulates) follows on the caller's stream.  (K9 per pair of views on a stream of its own under the next
    pair's K7 was measured in round 2 and lost 3-10 %: DESIGN §3.)
    Everything on the caller's stream when GDR_RENDER_SIDE=0 or for a single view.

        sides = _SideViews(dev, V, H, W)      # after all torch-side preparation: the side streams wait for this point
        for lo, n in sides.groups():
            ... K7 of views lo .. lo+n-1 on sides.stream(v) ...
            ... K9 of the group on sides.k9_stream(lo, n), accumulate = lo > 0 ...
        sides.join()
    """

    def __init__(self, dev, n, H, W):
        self.main = torch.cuda.current_stream()
        self.n = n
        self.side = self.k7_streams(dev, n, H, W, unique=True)
        if self.side:
            ready = torch.cuda.Event()
            ready.record(self.main)
            for sd in self.side:
                sd.wait_event(ready)

    @staticmethod
    def k7_streams(dev, n, H, W, unique=False):
        """The torch stream K7 of each of n views will run on, or None: caller's stream only.  unique: the distinct streams."""
        ns = side_count(H, W) if K.BWD_STREAMS is None else K.BWD_STREAMS
        if not (K.RENDER_SIDE and ns > 0 and n > 1):
            return None
        side = _view_streams(dev, min(ns, n))
        return side if unique else [side[k % len(side)] for k in range(n)]

    def groups(self):
        """(first view, views) of every K9 launch."""
        return [(lo, min(L.GDR_MAX_VIEWS, self.n - lo)) for lo in range(0, self.n, L.GDR_MAX_VIEWS)]

    def _side_of(self, k):
        return self.side[k % len(self.side)]

    def stream(self, k):
        if self.side:
            return C.c_void_p(self._side_of(k).cuda_stream)
        return C.c_void_p(self.main.cuda_stream)

    def k9_stream(self, lo, n):
        """The stream for K9 of views [lo, lo+n) (the caller's), made to wait for their K7."""
        if self.side:
            for sd in {self._side_of(k) for k in range(lo, lo + n)}:
                done = torch.cuda.Event()
                done.record(sd)
                self.main.wait_event(done)
        return C.c_void_p(self.main.cuda_stream)

    def join(self):
        if self.side:
            for sd in self.side:
                done = torch.cuda.Event()
                done.record(sd)
                self.main.wait_event(done)


class GroupMismatch(RuntimeError):
    """A later call of a render group was handed values that differ from the group's (viewgroup.py falls back to an
    independent node for that call)."""


def _view_opts():
    """The test / A-B switches of this module as the library's gdr_view_opts (-1 = the library's policy)."""
    return L.GdrViewOpts(-1 if K.SEG_LEN is None else max(0, int(K.SEG_LEN)) // 256 * 256,
                         -1 if K.DEEP_MAX_BUSY is None else max(0, int(K.DEEP_MAX_BUSY)),
                         -1 if K.DEEP_MIN_MEAN is None else max(0, int(K.DEEP_MIN_MEAN)),
                         int(K.FORCE_GLOBAL_SORT), int(K.FORCE_RADIX_PARTITION), int(not K.LAUNCH_HINTS))


def view_opts_tuple():
    """_view_opts() as the six ints the compiled boundary takes."""
    return (-1 if K.SEG_LEN is None else max(0, int(K.SEG_LEN)) // 256 * 256,
            -1 if K.DEEP_MAX_BUSY is None else max(0, int(K.DEEP_MAX_BUSY)),
            -1 if K.DEEP_MIN_MEAN is None else max(0, int(K.DEEP_MIN_MEAN)),
            int(K.FORCE_GLOBAL_SORT), int(K.FORCE_RADIX_PARTITION), int(not K.LAUNCH_HINTS))


def forward_view_native(call, s, inp, N, H, W, surfel, out, same_as, dev, stream):
    """ONE native call per view (include/gdr.h gdr_forward_view / gsr_forward_view, round 4): the library carves one
    allocation, runs K1, binning and K6 sized by the device counter, reads the duplicate count back through its pooled
    pinned buffer and keeps the per-shape history the next call is planned from.  Returns (workspace tensor,
    GdrViewState).  A GDR_ERR_WORKSPACE answer (first call of a shape, or a scene that outgrew the slack) is followed by
    an exactly planned second call."""
    lib = L.load()
    opts, plan, vs = _view_opts(), L.GdrViewPlan(), L.GdrViewState()
    same = None
    if same_as:
        same = L.GdrSameAs()
        same.n = len(same_as)
        for k, (t, ref) in enumerate(same_as):
            same.a[k], same.b[k], same.n_bytes[k] = t.data_ptr(), ref.data_ptr(), t.numel() * 4
        same = C.byref(same)
    exact = 0
    for _ in range(4):
        L.check(lib.gdr_view_plan_for(N, H, W, int(surfel), exact, C.byref(opts), C.byref(plan)), "gdr_view_plan_for")
        if not K.DEFER_D and not exact:     # upstream's flow: the count is read back before anything is sized
            plan.have_binning = 0
        ws = torch.empty(max(int(plan.bytes), 256), dtype=torch.uint8, device=dev)
        rc = call(s, inp, C.byref(plan), C.c_void_p(ws.data_ptr()), C.byref(opts), same, out, C.byref(vs), stream)
        if rc == L.GDR_OK:
            return ws, vs
        if rc != L.GDR_ERR_WORKSPACE:
            L.check(rc, "gdr_forward_view")
        exact = max(1, int(vs.D))
    raise RuntimeError("gdr_forward_view: the duplicate count kept growing between calls")


def forward_raw(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, same_as=None):
    """Un-differentiated forward. Returns (color, radii, depth, alpha, state, keep).
    same_as: optional [(tensor, reference tensor)] pairs a render group wants verified equal bit for bit (viewgroup.py):
    compared on the device next to K1, the verdict travels with the duplicate count (the spare word behind
    gdr_geom.num_rendered, same 8-byte copy) — no extra copy, no extra synchronisation; a difference raises GroupMismatch."""
    lib = L.load()
    _require_hip(means3D, "means3D")
    dev = means3D.device
    means3D = _f32(means3D, dev)
    opacities = _f32(opacities, dev)
    sh = _f32(sh, dev)
    colors_precomp = _f32(colors_precomp, dev)
    scales = _f32(scales, dev)
    rotations = _f32(rotations, dev)
    cov3Ds_precomp = _f32(cov3Ds_precomp, dev)
    N = int(means3D.shape[0])
    M = int(sh.shape[1]) if sh.numel() else 0
    H, W = int(raster_settings.image_height), int(raster_settings.image_width)
    if opacities.numel() != N:
        raise RuntimeError("opacities must have N elements")
    keep = [means3D, opacities, sh, colors_precomp, scales, rotations, cov3Ds_precomp]
    with torch.cuda.device(dev):
        s = _settings_struct(raster_settings, dev, keep)
        inp = _inputs_struct(N, M, means3D, opacities, sh, colors_precomp, scales, rotations, cov3Ds_precomp)
        f32 = dict(dtype=torch.float32, device=dev)
        color = torch.empty(3, H, W, **f32)
        depth = torch.empty(1, H, W, **f32)
        alpha = torch.empty(1, H, W, **f32)
        radii = torch.empty(N, dtype=torch.int32, device=dev)
        out = L.GdrOutputs(color.data_ptr(), depth.data_ptr(), alpha.data_ptr(), _ptr(radii))
        ws, vs = forward_view_native(lib.gdr_forward_view, C.byref(s), C.byref(inp), N, H, W, False, C.byref(out), same_as,
                                     dev, _stream())
        st = _State()
        st.N, st.M, st.H, st.W, st.D = N, M, H, W, int(vs.D)
        st.geom_buf = st.bin_buf = st.img_buf = ws      # one allocation, carved by the library
        st.view, st.geom, st.bin, st.img = vs, vs.geom, vs.bin, vs.img
        keep.append(s)      # (the backward of this call reuses the struct: its device tensors are in `keep` already)
        if same_as and vs.differ:
            raise GroupMismatch(
                "diff_gaussian_rasterization: this call's opacities / scales / rotations have the autograd provenance of "
                "an earlier call's (same ops on the same sources) but different values — a source tensor was modified in "
                "place between the calls, outside autograd's view.")
    return color, radii, depth, alpha, st, keep


def backward_raw(st: _State, keep, raster_settings, radii, grad_color, grad_depth, grad_alpha):
    """Returns dict of gradients (all fp32, on the inputs' device)."""
    lib = L.load()
    means3D, opacities, sh, colors_precomp, scales, rotations, cov3Ds_precomp = keep[:7]
    dev = means3D.device
    N, M = st.N, st.M
    f32 = dict(dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        keep2: list = []
        s = keep[-1] if isinstance(keep[-1], L.GdrSettings) else _settings_struct(raster_settings, dev, keep2)
        inp = _inputs_struct(N, M, means3D, opacities, sh, colors_precomp, scales, rotations, cov3Ds_precomp)
        gc = _f32(grad_color, dev)
        gd = None if grad_depth is None else _f32(grad_depth, dev)
        ga = None if grad_alpha is None else _f32(grad_alpha, dev)
        use_sh, use_cov = sh.numel() > 0, cov3Ds_precomp.numel() > 0
        g = dict(
            means3D=torch.empty(N, 3, **f32), means2D=torch.empty(N, 4, **f32),
            shs=torch.empty(N, M, 3, **f32) if use_sh else None,
            colors_precomp=None if use_sh else torch.empty(N, 3, **f32),
            opacities=torch.empty(N, 1, **f32),
            scales=None if use_cov else torch.empty(N, 3, **f32),
            rotations=None if use_cov else torch.empty(N, 4, **f32),
            cov3D_precomp=torch.empty(N, 6, **f32) if use_cov else None)
        scratch = torch.empty(max(N, 1) * 16, **f32)
        gin = L.GdrGradInputs(gc.data_ptr(), _ptr(gd), _ptr(ga))
        gout = L.GdrGradOutputs(_ptr(g["means3D"]), _ptr(g["means2D"]), _ptr(g["shs"]),
                                _ptr(g["colors_precomp"]), _ptr(g["opacities"]), _ptr(g["scales"]),
                                _ptr(g["rotations"]), _ptr(g["cov3D_precomp"]), scratch.data_ptr(), 0, 0)
        L.check(lib.gdr_backward(C.byref(s), C.byref(inp), C.byref(st.geom), C.byref(st.bin),
                                 C.byref(st.img), st.D, _ptr(radii), C.byref(gin), C.byref(gout),
                                 _stream()), "gdr_backward")
    return g


def _early_records_begin(ctx, dev, V, N, H, W, floats):
    """Gradient records of the V views (V, N * floats): allocated at the START of the forward, and the streams their K7 will
    run on (the first side streams, which also carry forward chains) are made to wait for THIS point of the caller's stream.
    _early_records_clear then zero-fills them on those streams behind the chains queued there — under the tail of the
    forward (the other views' K6), not in front of every K7 (64 B per Gaussian and view: 130 us of HBM time per C4 step).
    The event has to be recorded before the caller's stream joins the forward's side streams: recorded after the join (as
    until round 3) the clears could only start once the whole forward was over (kernel timeline of a C4 step: 90 us of
    fills between the last K6 and the first K7 with nothing else running).
    Only with side streams and only if a gradient was asked for; ctx.recs is consumed by the first backward (a second one
    clears its own).  Returns what _early_records_clear needs."""
    ctx.recs = None
    if not (K.EARLY_CLEAR and N > 0 and any(ctx.needs_input_grad[:6])):
        return None
    streams = _SideViews.k7_streams(dev, V, H, W)
    if streams is None:
        return None
    recs = torch.empty(V, N * floats, dtype=torch.float32, device=dev)
    # `recs` comes from the caller's stream's allocator pool: the block may still be in use by kernels queued on that
    # stream, and not every K7 stream is one of the forward's streams — each K7 stream waits for this point of the
    # caller's stream before it touches the block, and the allocator is told about every stream that uses it
    allocated = torch.cuda.Event()
    allocated.record(torch.cuda.current_stream())
    for sd in set(streams):
        sd.wait_event(allocated)
        recs.record_stream(sd)
    return recs, streams


def _early_records_clear(ctx, pending):
    if pending is None:
        return
    recs, streams = pending
    lib, row = L.load(), recs.shape[1] * 4
    distinct = list(dict.fromkeys(streams))
    if len(distinct) == 1 or k7_views_mode(0, 0, 0):
        # one-launch K7 (the default) runs on the caller's stream behind _join_record_clears: ONE zero-fill of the whole
        # (V, N * floats) block on the first K7 side stream instead of a call per view (5.7 us of host time each)
        L.check(lib.gdr_clear_async(C.c_void_p(recs.data_ptr()), row * recs.shape[0], C.c_void_p(distinct[0].cuda_stream)),
                "gdr_clear_async")
        distinct = distinct[:1]
    else:
        for v in range(recs.shape[0]):   # (a C call per view on the raw stream handle: a stream context + zero_() cost 25 us each)
            L.check(lib.gdr_clear_async(C.c_void_p(recs.data_ptr() + v * row), row, C.c_void_p(streams[v].cuda_stream)),
                    "gdr_clear_async")
    ctx.recs = recs
    ctx.recs_streams = distinct    # (whatever K7 mode the backward runs in, it joins these first: _join_record_clears)


def _early_records(ctx, dev, V, N, H, W, floats):
    """Both steps at the end of a forward (callers whose forward is not split yet)."""
    _early_records_clear(ctx, _early_records_begin(ctx, dev, V, N, H, W, floats))


def _join_record_clears(ctx):
    """The caller's stream waits for the early clears queued on the K7 side streams (one-launch K7: it runs on the caller's)."""
    streams = getattr(ctx, "recs_streams", None)
    if streams and getattr(ctx, "recs", None) is not None:
        main = torch.cuda.current_stream()
        for sd in streams:
            ev = torch.cuda.Event()
            ev.record(sd)
            main.wait_event(ev)
    ctx.recs_streams = None


def _view_arrays(states, lo, n, cleared):
    """ctypes arrays (geoms, binnings, images) of views [lo, lo + n) for the *_views entry points."""
    g_arr, b_arr, i_arr = (L.GdrGeom * n)(), (L.GdrBinning * n)(), (L.GdrImage * n)()
    for k in range(n):
        st = states[lo + k]
        st.bin.grad_rec_cleared = cleared
        g_arr[k], b_arr[k], i_arr[k] = st.geom, st.bin, st.img
        g_arr[k].cov3D = states[0].geom.cov3D
    return g_arr, b_arr, i_arr


def _take_records(ctx, lo, n, N, floats, dev):
    """(records of views [lo, lo + n), already cleared?) for one K9 group of a backward."""
    if getattr(ctx, "recs", None) is not None and lo + n <= ctx.recs.shape[0]:
        recs = ctx.recs[lo:lo + n]
        if lo + n == ctx.recs.shape[0]:
            ctx.recs = None
        return recs, 1
    return torch.empty(n, max(N, 1) * floats, dtype=torch.float32, device=dev), 0


# ---- gradient sinks: K9 writes a leaf's gradient where the collective needs it ------------------------------------------
# The reference's DDP reduces gradients in place (/root/reference/train_lightning.py:74).  View-sharded rendering sums the
# packed per-Gaussian gradients over the ranks (multiview.allreduce_gaussian_grads: reduce-scatter + all-gather of ONE packed
# buffer); until round 4 the gradients came out of K9 in fresh tensors and were copied into that buffer first (472 MB at 2 M
# Gaussians: 0.37 ms of a 3 ms step before a byte crossed xGMI).  A registered sink = "the gradient of THIS leaf belongs at
# THIS address": the multi-view nodes hand the address to K9 as its output pointer (gdr_preprocess_backward_views takes
# caller-provided outputs) and return an alias of it, which autograd's AccumulateGrad adopts as `.grad` without a copy.
# Only when that is safe: the node's input IS the registered leaf (same memory, fp32), the leaf has no gradient yet (an
# existing `.grad` is accumulated into in place — it may be this very memory), and no other node of the running backward
# pass has taken the sink.
_GRAD_SINKS: dict = {}      # data_ptr of the leaf -> [weakref(leaf), sink tensor (the leaf's shape), graph task that took it]


def register_grad_sink(leaf: torch.Tensor, sink: torch.Tensor):
    import weakref
    if sink.shape != leaf.shape or sink.dtype != torch.float32 or leaf.dtype != torch.float32 or sink.device != leaf.device \
            or not sink.is_contiguous():
        raise ValueError("register_grad_sink: the sink must be a contiguous fp32 tensor of the leaf's shape on its device")
    with _HIST_LOCK:
        for k in [k for k, e in _GRAD_SINKS.items() if e[0]() is None]:
            del _GRAD_SINKS[k]
        _GRAD_SINKS[leaf.data_ptr()] = [weakref.ref(leaf), sink, None]


def unregister_grad_sinks(sinks=None):
    """Forget every registered sink, or only those whose sink tensor is one of `sinks` (a packed buffer that is being
    dropped: multiview._grad_pack's eviction) — the registry holds the sink tensors strongly."""
    with _HIST_LOCK:
        if sinks is None:
            _GRAD_SINKS.clear()
            return
        ptrs = {t.data_ptr() for t in sinks}
        for k in [k for k, e in _GRAD_SINKS.items() if e[1].data_ptr() in ptrs or e[0]() is None]:
            del _GRAD_SINKS[k]


def _sink_for(t: torch.Tensor):
    """The registered sink for a node input `t`, or None (see above)."""
    if not _GRAD_SINKS:
        return None
    e = _GRAD_SINKS.get(t.data_ptr())
    if e is None:
        return None
    leaf = e[0]()
    # `t` must BE the registered leaf (round-5 advisor finding): a view of it with another shape (p.view(N, 1), p[:, None]) or a
    # second leaf on the same memory (p.detach().requires_grad_()) has the leaf's address and element count, but a sink alias of
    # the leaf's shape is the wrong gradient for the first and somebody else's buffer for the second.  Saved inputs unpack to the
    # caller's own tensor object, so identity is the test; anything else gets an ordinary buffer.
    if (leaf is None or t is not leaf or t.grad_fn is not None or leaf.grad is not None or not leaf.is_leaf or not leaf.requires_grad
            or t.shape != leaf.shape or t.dtype != torch.float32 or leaf._backward_hooks):
        return None
    task = torch._C._current_graph_task_id()
    with _HIST_LOCK:
        if e[2] is not None and e[2] == task:
            return None            # another node of this pass already writes there: this one gets its own buffer
        e[2] = task
    return e[1]


def _grad_buffers(N, M, f32, inputs=None, scale_cols=3):
    """Output buffers of one K8+K9 launch.  inputs: the node's (means3D, sh, opacities, scales, rotations) as saved for
    backward — an input with a registered gradient sink gets the sink as its buffer.  Returns (buffers, keys backed by a sink)."""
    shapes = dict(means3D=(N, 3), means2D=(N, 4), shs=(N, M, 3), opacities=(N, 1), scales=(N, scale_cols), rotations=(N, 4))
    g, sunk = {}, set()
    if inputs is not None and _GRAD_SINKS:
        for k, t in zip(("means3D", "shs", "opacities", "scales", "rotations"), inputs):
            s = _sink_for(t)
            if s is not None and s.numel() == math.prod(shapes[k]):
                g[k] = s
                sunk.add(k)
    for k, shp in shapes.items():
        if k not in g:
            g[k] = torch.empty(*shp, **f32)
    return g, sunk


def _returned(g, sunk, key, dtype):
    """What a node returns for gradient `key`: a sink-backed buffer goes back as a NEW alias of the sink (AccumulateGrad adopts
    an incoming gradient only if nobody else holds the tensor object; the registry does hold the sink's)."""
    t = g[key]
    if key in sunk:
        return t.detach()
    return t if t.dtype == dtype else t.to(dtype)


def _kept_settings(ctx, dev, keep2):
    """The V settings structs of a multi-view node's forward (kept behind its device tensors), or fresh ones."""
    arr = ctx.keep_rest[-1] if ctx.keep_rest else None
    if isinstance(arr, C.Array) and len(arr) == len(ctx.settings_list) and isinstance(arr[0], L.GdrSettings):
        return list(arr)
    return [_settings_struct(rs, dev, keep2) for rs in ctx.settings_list]


def _save_inputs(ctx, keep, n=7):
    """The n f32 input tensors of a node go through ctx.save_for_backward, as in the upstream extension: autograd's
    version counters then turn an in-place update between forward and backward (an optimizer step, a densification
    write) into an error instead of K9 silently differentiating the modified values, and the buffers are released with
    the graph.  The rest of `keep` (device copies of the settings that side-stream kernels may still read, ints) only
    has to stay alive."""
    ctx.save_for_backward(*keep[:n])
    ctx.keep_rest = list(keep[n:])


def _saved_inputs(ctx):
    return list(ctx.saved_tensors) + ctx.keep_rest


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                cov3Ds_precomp, raster_settings):
        color, radii, depth, alpha, st, keep = forward_raw(
            means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings)
        ctx.raster_settings = raster_settings
        ctx.state = st
        ctx.pace = viewgroup.pace()
        _save_inputs(ctx, keep)
        ctx.radii = radii
        ctx.means2D_shape = tuple(means2D.shape)
        ctx.in_dtypes = tuple(t.dtype for t in (means3D, means2D, sh, colors_precomp, opacities, scales,
                                                rotations, cov3Ds_precomp))
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        viewgroup.note_backward(ctx.pace)
        g = backward_raw(ctx.state, _saved_inputs(ctx), ctx.raster_settings, ctx.radii, grad_color, grad_depth,
                         grad_alpha)
        gm2 = g["means2D"]
        cols = ctx.means2D_shape[1] if len(ctx.means2D_shape) == 2 else 4
        if cols == 4:
            pass
        elif cols == 3:  # legacy caller (point_decoder/layers/gaussian_renderer.py): xy signed, z = 0
            gm2 = torch.cat([gm2[:, :2], torch.zeros_like(gm2[:, :1])], dim=1)
        else:
            gm2 = gm2[:, :cols].contiguous()
        grads = [g["means3D"], gm2, g["shs"], g["colors_precomp"], g["opacities"], g["scales"],
                 g["rotations"], g["cov3D_precomp"]]
        grads = [None if t is None else (t if t.dtype == dt else t.to(dt)) for t, dt in zip(grads, ctx.in_dtypes)]
        return (*grads, None)


# --------------------------------------------------------------------------------------------
# Multi-view fused entry point.  One autograd node renders V views of
# ONE Gaussian set: the adaptor's activations (sigmoid / exp / normalize, renderer.py:225-230)
# run inside K1/K9, the V values of num_rendered are read back with ONE host sync, and the
# per-Gaussian gradients are summed over the views inside K9 (accumulate mode) instead of V
# separate autograd accumulation passes.  Same arithmetic as V calls of rasterize_gaussians.
# --------------------------------------------------------------------------------------------
RAW_ALL = L.GDR_IN_RAW_OPACITY | L.GDR_IN_RAW_SCALES | L.GDR_IN_RAW_ROTATIONS

# ---- the duplicate count D without a host stall --------------------------------------------------------------
# K1 leaves D (the reference's num_rendered) on the device; the sort buffers are sized by it, and upstream reads it back
# in the middle of every forward call — the host waits for K1, the GPU then waits for the host to allocate and launch.
# Here the FIRST call of a shape (N, H, W, V) does the same.  Later calls carve the binning workspaces for a CAPACITY
# derived from the counts this shape has produced so far, point the binning kernels at the device word
# (gdr_binning.d_dev: a device-sized call, include/gdr.h) and enqueue binning + K6 of every view right behind K1; the
# counts travel to pinned host memory meanwhile and are compared with the capacity once everything is enqueued.  A view
# that did not fit (nothing was written out of bounds, but its lists are truncated) is repeated with an exactly sized
# workspace before the call returns, so the results never depend on the guess.
D_SLACK = 1.5   # capacity = slack x the largest recent count of the shape (measured:
# 1.02 / 1.25 / 2.0 run at the same speed — surplus workgroups leave at once —, so the slack only costs memory)
_D_HINT: dict = {}   # shape key -> decaying maximum of duplicates PER GAUSSIAN of one view of that shape
_HIST_LOCK = threading.Lock()   # the per-shape histories (_D_HINT, _LAUNCH_STATS, _HINT_STATE, _side_streams) are process-wide:
# two host threads driving distinct workspaces (include/gdr.h: the library below is thread-safe for those) may share them


def shape_key(N, *rest):
    """Key of the per-shape histories (duplicate counts, launch reports): the Gaussian count enters as its power-of-two
    bucket — a densifying model renders a different N every step, and what carries over between neighbouring N is the
    number of duplicates per Gaussian, not the number of duplicates."""
    return (int(N).bit_length(),) + tuple(rest)


def _d_capacity(key, N):
    """Entries to carve each view's binning workspace for, or None: no history yet (or GDR_DEFER_D=0) -> read D back."""
    with _HIST_LOCK:
        h = _D_HINT.get(key) if K.DEFER_D else None
    return None if h is None else int(h * N * D_SLACK) + 4096


def _d_record(key, d_host, N):
    with _HIST_LOCK:
        prev = _D_HINT.pop(key, 0.0)     # (re-inserted at the end: the dict doubles as an LRU of 64 shapes)
        _D_HINT[key] = max(max(d_host, default=0) / max(N, 1), prev * 0.97)
        if len(_D_HINT) > 64:
            _D_HINT.pop(next(iter(_D_HINT)))


class _CountReadback:
    """The V duplicate counters on their way to pinned host memory, behind K1, on `stream` (default: the caller's).
    A multi-view node puts the copy in front of its last forward chain (that stream waits for K1 anyway) instead of in
    front of the caller's: same-box A/B C2 +1.2 %, C3 +0.9 %, C4 +0.5 %.
    The copy and its event are two calls into the library (gdr_host_copy_begin / _wait, pooled events) into a pooled pinned
    buffer: a pinned tensor + copy_ + torch.cuda.Event per call cost 21 us of host time (scripts/host_split2.py)."""
    _pool: dict = {}     # numel -> [pinned int32 tensors not in flight]

    def __init__(self, counters, stream=None):
        self.counters, self.ticket = counters, None
        n = counters.numel()
        with _HIST_LOCK:
            free = self._pool.get(n)
            self.host = free.pop() if free else None
        if self.host is None:
            try:
                self.host = torch.empty(n, dtype=torch.int32, pin_memory=True)
            except RuntimeError:    # no page-locked memory: a blocking copy when the counts are needed
                return
        raw = stream.cuda_stream if stream is not None else _raw_stream(counters.device.index)
        ticket = C.c_void_p()
        L.check(L.load().gdr_host_copy_begin(self.host.data_ptr(), counters.data_ptr(), 4 * n, C.c_void_p(raw),
                                             C.byref(ticket)), "gdr_host_copy_begin")
        self.ticket = ticket

    def wait(self):
        if self.host is None:
            return [int(d) & 0xFFFFFFFF for d in self.counters.cpu().tolist()]
        ticket, self.ticket = self.ticket, None
        L.check(L.load().gdr_host_copy_wait(ticket), "gdr_host_copy_wait")
        vals = [int(d) & 0xFFFFFFFF for d in self.host.tolist()]
        with _HIST_LOCK:
            self._pool.setdefault(self.host.numel(), []).append(self.host)
        self.host = None
        return vals

    def __del__(self):   # never waited for (an exception in between): the ticket and the buffer go back once the copy is done
        if getattr(self, "ticket", None) is not None:
            try:
                L.load().gdr_host_copy_wait(self.ticket)
            except Exception:
                pass


# ---- launch-size feedback (gdr_binning.stats_out / hint_*) ----------------------------------------------------
# The binning stage of a view reports how many tiles fell into the tile sort's long / medium class and whether the deep
# forward applied; the next call of the same shape sizes those launches from it (empty classes: 1 workgroup instead of
# 256 / 512 with 144 / 40 KB of LDS each; no deep launch of 3072 workgroups that leave at once).  Results never depend
# on it: the classes walk their tiles with a grid stride, and K6 renders every tile the standard way without the deep
# launch.  The words live in pinned host memory the kernels write directly (4 words per view and shape, kept for the
# life of the process: the GPU may still be writing when a shape is last used).
_LAUNCH_STATS: dict = {}


_HINT_STATE: dict = {}   # shape key -> [long tiles, medium tiles, calls until the deep launch may be dropped]: decaying maxima


def _launch_stats(key, V):
    """(pinned int32 (V, 4) the kernels of this call report into — {long tiles, medium tiles, deep flag, busy tiles} per
    view, -1 = no call yet —, hints for this call = (workgroups of the long class, of the medium class, no deep launch,
    busy tiles) or None).  The hints are the decaying maximum over the views and the recent calls of the shape (cameras change from
    step to step) plus 25 %, and never below 16 / 32 workgroups: a scene that suddenly has a hundred long lists costs a
    few rounds on a small grid, not one workgroup sorting them all."""
    if not K.LAUNCH_HINTS:
        return None, None
    key = (torch.cuda.current_device(),) + tuple(key)   # one report tensor per device and shape
    with _HIST_LOCK:
        t = _LAUNCH_STATS.get(key)
        if t is None:
            if len(_LAUNCH_STATS) >= 1024:
                return None, None
            try:
                t = torch.full((V, 4), -1, dtype=torch.int32).pin_memory()
            except RuntimeError:    # no page-locked memory to be had: the feedback is an optimisation, not a requirement
                return None, None
            _LAUNCH_STATS[key] = t
        seen = [r for r in t.tolist() if r[0] >= 0]
        if not seen:
            return t, None
        n_long, n_medium, deep = max(r[0] for r in seen), max(r[1] for r in seen), any(r[2] for r in seen)
        st = _HINT_STATE.setdefault(key, [0, 0, 0])
        st[0], st[1] = max(n_long, st[0] * 9 // 10), max(n_medium, st[1] * 9 // 10)
        st[2] = 8 if deep else max(0, st[2] - 1)
        st = list(st)
    # (long class: -1 = "no list beyond the medium class in the recent calls" -> the 144 KB launch is skipped, binning.hip)
    return t, (-1 if st[0] == 0 else max(16, st[0] + st[0] // 4 + 1), max(32, st[1] + st[1] // 4 + 1), int(st[2] == 0),
               max(r[3] for r in seen))


def _carve_binning(lib, st, entries, tiles, d_dev=None, stats=None, hints=None):
    """Workspace of one view for `entries` duplicates (exact count, or a capacity with d_dev = the device counter);
    stats / hints: this view's row of the report tensor and the call's launch hints (_launch_stats)."""
    seg_len = _seg_len_for(entries if d_dev is None else int(entries / D_SLACK), tiles, None if hints is None else hints[3])
    need = lib.gdr_binning_bytes_for(entries, seg_len, st.N, tiles)
    if st.bin_buf is None or st.bin_buf.numel() < need:
        st.bin_buf = torch.empty(need, dtype=torch.uint8, device=st.geom_buf.device)
    L.check(lib.gdr_binning_carve_for(st.bin_buf.data_ptr(), entries, seg_len, st.N, tiles, C.byref(st.bin)),
            "gdr_binning_carve_for")
    if K.FORCE_RADIX_PARTITION:
        st.bin.tile_hist, st.bin.hist_width = None, 0
    st.bin.global_sort = int(K.FORCE_GLOBAL_SORT)
    st.bin.d_dev = d_dev
    st.D = entries
    if K.DEEP_MAX_BUSY is not None:
        st.bin.deep_max_busy = max(0, int(K.DEEP_MAX_BUSY))
    if K.DEEP_MIN_MEAN is not None:
        st.bin.deep_min_mean = max(0, int(K.DEEP_MIN_MEAN))
    if hints is not None:
        st.bin.hint_long, st.bin.hint_medium, st.bin.hint_no_deep = hints[:3]
    if stats is not None:
        st.bin.stats_out = stats.data_ptr()


def _forward_views_impl(means3D, means2D, sh, opacities, scales, rotations, settings_list, flags, loss_spec=None):
    """The forward of a multi-view node in ONE native call (include/gdr.h gdr_forward_views, round 4): K1 for all views (one
    launch per <= 8 views), then every view's chain binning -> K6 on one of K.FWD_STREAMS streams, the caller's included; the
    duplicate counts are read back once, after everything is enqueued; one allocation for all views, carved by the library.
    (Until round 3 this was ~15 ABI calls per view from Python: at reference-scale scenes the chains of the four streams started
    60-100 us apart, the host being the slower side — kernel timeline of a C2 step, profiles/r04_timeline_c2.txt.)
    Returns (colors, radii, depths, alphas, states, keep, in_dtypes)."""
    lib = L.load()
    _require_hip(means3D, "means3D")
    dev = means3D.device
    in_dtypes = tuple(t.dtype for t in (means3D, means2D, sh, opacities, scales, rotations))
    means3D, sh = _f32(means3D, dev), _f32(sh, dev)
    opacities, scales, rotations = _f32(opacities, dev), _f32(scales, dev), _f32(rotations, dev)
    N, M, V = int(means3D.shape[0]), int(sh.shape[1]), len(settings_list)
    H, W = int(settings_list[0].image_height), int(settings_list[0].image_width)
    if any(int(rs.image_height) != H or int(rs.image_width) != W for rs in settings_list):
        raise RuntimeError("render_views: all views must share one image size")
    if V > L.GDR_MAX_NODE_VIEWS:
        raise RuntimeError(f"render_views: {V} views in one node; one native forward call takes at most {L.GDR_MAX_NODE_VIEWS} "
                           "(include/gdr.h GDR_MAX_NODE_VIEWS) — split the view list")
    e = empty_f32(dev)
    keep = [means3D, opacities, sh, e, scales, rotations, e]
    f32 = dict(dtype=torch.float32, device=dev)
    # one tensor per view (not slices of a stacked buffer): a per-view loss then back-propagates straight into
    # that view's gradient, without autograd's select-backward zero-fill + add of the whole stack per view
    lossgrad = loss_spec is not None and loss_spec[0] == "lossgrad"   # abs-grad-only path: no image leaves K6
    colors = [torch.empty(3, H, W, **f32) for _ in range(V)]    # (lossgrad: d loss / d colour per pixel instead)
    depths = [None if lossgrad else torch.empty(1, H, W, **f32) for _ in range(V)]
    alphas = [None if lossgrad else torch.empty(1, H, W, **f32) for _ in range(V)]
    radii = torch.empty(V, N, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        main = torch.cuda.current_stream()
        inp = _inputs_struct(N, M, means3D, opacities, sh, e, scales, rotations, e, flags)
        s_arr = (L.GdrSettings * V)(*[_settings_struct(rs, dev, keep) for rs in settings_list])
        o_arr = (L.GdrOutputs * V)(*[L.GdrOutputs(colors[v].data_ptr(), _ptr(depths[v]), _ptr(alphas[v]), _ptr(radii[v]))
                                     for v in range(V)])
        mode_, tg_arr, wd_, wa_, gs_, ls_ = 0, None, 0.0, 0.0, 1.0, None
        if lossgrad:
            _, targets, go_scale, losses = loss_spec
            mode_, gs_, ls_ = 2, float(go_scale), losses
        elif loss_spec is not None:
            targets, w_depth, w_alpha, losses = loss_spec
            mode_, wd_, wa_, ls_ = 1, float(w_depth), float(w_alpha), losses
        if mode_:
            tg_arr = (C.c_void_p * V)(*[targets[v].data_ptr() for v in range(V)])
        # The views' chains round-robin over K.FWD_STREAMS streams, the caller's included: the binning is ~6 short,
        # latency-bound kernels per view that overlap the VALU-bound K6 of other views, and two concurrent K6 fill the CUs that
        # one view's skewed tile lists and kernel tails leave idle.  Four streams = the number of hardware queues a process
        # gets by default (more alias onto the same queues and serialise).
        nfs = max(1, min(K.FWD_STREAMS, V)) if K.RENDER_SIDE and V > 1 and side_count(H, W) > 0 else 1
        fstreams = [main] + _view_streams(dev, nfs - 1)
        st_arr = (C.c_void_p * nfs)(*[fs.cuda_stream for fs in fstreams])
        opts, plan, vs_arr = _view_opts(), L.GdrViewsPlan(), (L.GdrViewState * V)()
        exact, ws = 0, None
        for _ in range(4):
            L.check(lib.gdr_views_plan_for(V, N, H, W, exact, C.byref(opts), C.byref(plan)), "gdr_views_plan_for")
            if not K.DEFER_D and not exact:     # upstream's flow: the counts are read back before anything is sized
                plan.view.have_binning = 0
            # (no record_stream for the side streams: the call makes the caller's stream wait for every one of them before it
            # returns, and the backward's streams start from the caller's — every use of the block is ordered before
            # anything the caller's stream does later, which is all the caching allocator needs)
            ws = torch.empty(max(int(plan.bytes), 256), dtype=torch.uint8, device=dev)
            if mode_ and exact:
                ls_.zero_()          # (a repeated call accumulates its losses again)
            rc = lib.gdr_forward_views(V, s_arr, C.byref(inp), C.byref(plan), C.c_void_p(ws.data_ptr()), C.byref(opts), o_arr,
                                       mode_, tg_arr, wd_, wa_, gs_, None if ls_ is None else C.c_void_p(ls_.data_ptr()), st_arr,
                                       nfs, vs_arr)
            if rc == L.GDR_OK:
                break
            if rc != L.GDR_ERR_WORKSPACE:
                L.check(rc, "gdr_forward_views")
            exact = max(1, max(int(vs_arr[v].D) for v in range(V)))
        else:
            raise RuntimeError("gdr_forward_views: the duplicate counts kept growing between calls")
        states = []
        for v in range(V):
            st = _State()
            st.N, st.M, st.H, st.W, st.D = N, M, H, W, int(vs_arr[v].D)
            st.geom_buf = st.bin_buf = st.img_buf = ws
            st.view, st.geom, st.bin, st.img = vs_arr, vs_arr[v].geom, vs_arr[v].bin, vs_arr[v].img
            states.append(st)
    keep.append(s_arr)     # the V settings structs: the backward reuses them (their device tensors are in `keep` already)
    return colors, radii, depths, alphas, states, keep, in_dtypes


class _RenderViews(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, opacities, scales, rotations, settings_list, flags):
        rs0 = settings_list[0]
        pending = _early_records_begin(ctx, means3D.device, len(settings_list), int(means3D.shape[0]), int(rs0.image_height),
                                       int(rs0.image_width), 16)
        colors, radii, depths, alphas, states, keep, in_dtypes = _forward_views_impl(
            means3D, means2D, sh, opacities, scales, rotations, settings_list, flags)
        ctx.states, ctx.settings_list, ctx.flags = states, settings_list, flags
        _save_inputs(ctx, keep)
        ctx.radii, ctx.in_dtypes, ctx.means2D_shape = radii, in_dtypes, tuple(means2D.shape)
        ctx.mark_non_differentiable(radii)
        _early_records_clear(ctx, pending)
        return (radii, *colors, *depths, *alphas)

    @staticmethod
    def backward(ctx, g_radii, *g_views):
        lib = L.load()
        means3D, opacities, sh, e, scales, rotations, _ = _saved_inputs(ctx)[:7]
        dev = means3D.device
        states = ctx.states
        N, M, V = states[0].N, states[0].M, len(states)
        H, W = states[0].H, states[0].W
        g_colors, g_depths, g_alphas = g_views[:V], g_views[V:2 * V], g_views[2 * V:3 * V]
        f32 = dict(dtype=torch.float32, device=dev)
        g = inp = None      # (allocated behind the K7 launch)
        with torch.cuda.device(dev):
            keep2: list = []
            stream = _stream()
            grads_in = []
            for v in range(V):  # torch-side preparation stays on the caller's stream
                gc = (_f32(g_colors[v], dev) if g_colors[v] is not None
                      else torch.zeros(3, H, W, dtype=torch.float32, device=dev))
                gd = None if g_depths[v] is None else _f32(g_depths[v], dev)
                ga = None if g_alphas[v] is None else _f32(g_alphas[v], dev)
                keep2 += [gc, gd, ga]
                grads_in.append((gc, gd, ga))
            sets = _kept_settings(ctx, dev, keep2)
            mode = k7_views_mode(H, W, N) if V > 1 else 0
            _join_record_clears(ctx)      # (always: the forward chose ONE clear on one stream from the mode it saw; if the mode was
            #                                switched in between — the A/B tooling does — per-view K7 launches must still wait for it)
            sides = _SideViews(dev, 1 if mode else V, H, W)  # after every torch-side preparation (the side streams wait for this point)
            for lo in range(0, V, L.GDR_MAX_VIEWS):
                n = min(L.GDR_MAX_VIEWS, V - lo)
                recs, cleared = _take_records(ctx, lo, n, N, 16, dev)  # one 64-byte gradient record per Gaussian per view
                s_arr = (L.GdrSettings * n)(*sets[lo:lo + n])
                g_arr, b_arr, i_arr = _view_arrays(states, lo, n, cleared)
                if mode:    # K7 of the n views in one launch on the caller's stream
                    gin_arr = (L.GdrGradInputs * n)(*[L.GdrGradInputs(gc.data_ptr(), _ptr(gd), _ptr(ga))
                                                      for gc, gd, ga in grads_in[lo:lo + n]])
                    rec_ptrs = (C.c_void_p * n)(*[recs[k].data_ptr() for k in range(n)])
                    L.check(lib.gdr_render_backward_views(n, s_arr, N, g_arr, b_arr, i_arr, gin_arr, rec_ptrs, int(mode == 1),
                                                          stream), "gdr_render_backward_views")
                for k in range(0 if mode else n):
                    v = lo + k
                    st = states[v]
                    gc, gd, ga = grads_in[v]
                    gin = L.GdrGradInputs(gc.data_ptr(), _ptr(gd), _ptr(ga))
                    L.check(lib.gdr_render_backward(C.byref(s_arr[k]), N, C.byref(g_arr[k]), C.byref(st.bin),
                                                    C.byref(st.img), C.byref(gin), recs[k].data_ptr(), sides.stream(v)),
                            "gdr_render_backward")
                if g is None:
                    g, sunk = _grad_buffers(N, M, f32, (means3D, sh, opacities, scales, rotations) if all(
                        dt == torch.float32 for dt in ctx.in_dtypes) else None)
                    inp = _inputs_struct(N, M, means3D, opacities, sh, e, scales, rotations, e, ctx.flags)
                r_arr = (C.c_void_p * n)(*[ctx.radii[lo + k].data_ptr() if N else None for k in range(n)])
                rec_arr = (C.c_void_p * n)(*[recs[k].data_ptr() for k in range(n)])
                gout = L.GdrGradOutputs(_ptr(g["means3D"]), _ptr(g["means2D"]), _ptr(g["shs"]), None,
                                        _ptr(g["opacities"]), _ptr(g["scales"]), _ptr(g["rotations"]), None, None,
                                        1 if lo > 0 else 0, 0)
                L.check(lib.gdr_preprocess_backward_views(n, s_arr, C.byref(inp), g_arr, r_arr, rec_arr,
                                                          C.byref(gout), sides.k9_stream(lo, n)),
                        "gdr_preprocess_backward_views")
                keep2.append(recs)
            sides.join()
        gm2 = g["means2D"]
        cols = ctx.means2D_shape[1] if len(ctx.means2D_shape) == 2 else 4
        if cols == 3:
            gm2 = torch.cat([gm2[:, :2], torch.zeros_like(gm2[:, :1])], dim=1)
        elif cols != 4:
            gm2 = gm2[:, :cols].contiguous()
        g["means2D"] = gm2
        grads = [_returned(g, sunk, k, dt) for k, dt in zip(("means3D", "means2D", "shs", "opacities", "scales", "rotations"),
                                                            ctx.in_dtypes)]
        return (*grads, None, None)


class _RenderViewsLoss(torch.autograd.Function):
    """V views of one Gaussian set AND their image losses in one node: K6 accumulates
    loss_v = mean((clamp(color_v) - target_v)^2) + w_depth mean(depth_v) + w_alpha mean(alpha_v) in its epilogue, K7
    forms the per-pixel upstream gradients in its prologue — no loss kernels, no dL/dimage tensors, no autograd nodes
    between the rasterizer and the scalar losses.  Returns (losses (V,), radii (V,N))."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, opacities, scales, rotations, settings_list, flags, targets, w_depth, w_alpha):
        dev = means3D.device
        V = len(settings_list)
        targets = [t.to(device=dev, dtype=torch.float32).contiguous() for t in targets]
        losses = torch.zeros(V, dtype=torch.float32, device=dev)
        rs0 = settings_list[0]
        pending = _early_records_begin(ctx, dev, V, int(means3D.shape[0]), int(rs0.image_height), int(rs0.image_width), 16)
        colors, radii, depths, alphas, states, keep, in_dtypes = _forward_views_impl(
            means3D, means2D, sh, opacities, scales, rotations, settings_list, flags,
            loss_spec=(targets, w_depth, w_alpha, losses))
        ctx.states, ctx.settings_list, ctx.flags = states, settings_list, flags
        _save_inputs(ctx, keep)
        ctx.radii, ctx.in_dtypes, ctx.means2D_shape = radii, in_dtypes, tuple(means2D.shape)
        ctx.colors, ctx.targets, ctx.w = colors, targets, (float(w_depth), float(w_alpha))
        ctx.mark_non_differentiable(radii)
        _early_records_clear(ctx, pending)
        return losses, radii

    @staticmethod
    def backward(ctx, g_losses, g_radii):
        lib = L.load()
        means3D, opacities, sh, e, scales, rotations, _ = _saved_inputs(ctx)[:7]
        dev = means3D.device
        states = ctx.states
        N, M, V = states[0].N, states[0].M, len(states)
        f32 = dict(dtype=torch.float32, device=dev)
        g = inp = None      # (allocated behind the K7 launch: the GPU idles between the loss and K7, kernel timeline of a C2 step)
        go = g_losses.to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            keep2: list = []
            stream = _stream()
            sets = _kept_settings(ctx, dev, keep2)
            mode = k7_views_mode(states[0].H, states[0].W, N) if V > 1 else 0
            _join_record_clears(ctx)      # (always: see _RenderViews.backward)
            sides = _SideViews(dev, 1 if mode else V, states[0].H, states[0].W)  # after every torch-side preparation (the side streams wait for this point)
            for lo in range(0, V, L.GDR_MAX_VIEWS):
                n = min(L.GDR_MAX_VIEWS, V - lo)
                recs, cleared = _take_records(ctx, lo, n, N, 16, dev)
                s_arr = (L.GdrSettings * n)(*sets[lo:lo + n])
                g_arr, b_arr, i_arr = _view_arrays(states, lo, n, cleared)
                if mode:    # K7 of the n views in one launch on the caller's stream
                    col_ptrs = (C.c_void_p * n)(*[ctx.colors[lo + k].data_ptr() for k in range(n)])
                    tgt_ptrs = (C.c_void_p * n)(*[ctx.targets[lo + k].data_ptr() for k in range(n)])
                    rec_ptrs = (C.c_void_p * n)(*[recs[k].data_ptr() for k in range(n)])
                    L.check(lib.gdr_render_backward_loss_views(n, s_arr, N, g_arr, b_arr, i_arr, col_ptrs, tgt_ptrs, ctx.w[0],
                                                               ctx.w[1], go[lo:lo + n].data_ptr(), rec_ptrs, int(mode == 1),
                                                               stream), "gdr_render_backward_loss_views")
                for k in range(0 if mode else n):
                    v = lo + k
                    st = states[v]
                    L.check(lib.gdr_render_backward_loss(C.byref(s_arr[k]), N, C.byref(g_arr[k]), C.byref(st.bin),
                                                         C.byref(st.img), ctx.colors[v].data_ptr(), ctx.targets[v].data_ptr(),
                                                         ctx.w[0], ctx.w[1], go[v:v + 1].data_ptr(), recs[k].data_ptr(),
                                                         sides.stream(v)), "gdr_render_backward_loss")
                if g is None:
                    g, sunk = _grad_buffers(N, M, f32, (means3D, sh, opacities, scales, rotations) if all(
                        dt == torch.float32 for dt in ctx.in_dtypes) else None)
                    inp = _inputs_struct(N, M, means3D, opacities, sh, e, scales, rotations, e, ctx.flags)
                r_arr = (C.c_void_p * n)(*[ctx.radii[lo + k].data_ptr() if N else None for k in range(n)])
                rec_arr = (C.c_void_p * n)(*[recs[k].data_ptr() for k in range(n)])
                gout = L.GdrGradOutputs(_ptr(g["means3D"]), _ptr(g["means2D"]), _ptr(g["shs"]), None,
                                        _ptr(g["opacities"]), _ptr(g["scales"]), _ptr(g["rotations"]), None, None,
                                        1 if lo > 0 else 0, 0)
                L.check(lib.gdr_preprocess_backward_views(n, s_arr, C.byref(inp), g_arr, r_arr, rec_arr,
                                                          C.byref(gout), sides.k9_stream(lo, n)),
                        "gdr_preprocess_backward_views")
                keep2.append(recs)
            sides.join()
        gm2 = g["means2D"]
        cols = ctx.means2D_shape[1] if len(ctx.means2D_shape) == 2 else 4
        if cols == 3:
            gm2 = torch.cat([gm2[:, :2], torch.zeros_like(gm2[:, :1])], dim=1)
        elif cols != 4:
            gm2 = gm2[:, :cols].contiguous()
        g["means2D"] = gm2
        grads = [_returned(g, sunk, k, dt) for k, dt in zip(("means3D", "means2D", "shs", "opacities", "scales", "rotations"),
                                                            ctx.in_dtypes)]
        return (*grads, None, None, None, None, None)


def _size_groups(settings_list):
    """Views grouped by image size, in first-appearance order: [(indices, settings)], one multi-view node each."""
    groups: dict = {}
    for v, rs in enumerate(settings_list):
        groups.setdefault((int(rs.image_height), int(rs.image_width)), []).append(v)
    return [(idx, [settings_list[v] for v in idx]) for idx in groups.values()]


def render_views_loss_raw(means3D, means2D, sh, opacities, scales, rotations, settings_list, targets_chw, w_depth=0.1,
                          w_alpha=0.1, flags=RAW_ALL):
    """Per-view losses (V,) of V views of one Gaussian set with the loss folded into K6/K7; targets_chw: V tensors
    (3,H,W).  Returns (losses, radii).  Views of different image sizes are rendered by one node per size."""
    groups = _size_groups(settings_list)
    if len(groups) <= 1:
        return _RenderViewsLoss.apply(means3D, means2D, sh, opacities, scales, rotations, list(settings_list), int(flags),
                                      list(targets_chw), float(w_depth), float(w_alpha))
    V = len(settings_list)
    losses, radii = [None] * V, [None] * V
    for idx, sets in groups:
        l, r = _RenderViewsLoss.apply(means3D, means2D, sh, opacities, scales, rotations, sets, int(flags),
                                      [targets_chw[v] for v in idx], float(w_depth), float(w_alpha))
        for k, v in enumerate(idx):
            losses[v], radii[v] = l[k], r[k]
    return torch.stack(losses), torch.stack(radii)


def render_views_raw(means3D, means2D, sh, opacities, scales, rotations, settings_list, flags=RAW_ALL):
    """V views of one Gaussian set in one autograd node.  With flags=RAW_ALL the opacity /
    scale / rotation tensors are the adaptor's RAW (pre-activation) tensors.
    Returns (colors, radii (V,N) int32, depths, alphas) with colors / depths / alphas LISTS of V per-view
    tensors (3,H,W) / (1,H,W) / (1,H,W).  Views of different image sizes (the multi-view kernels need one size per
    launch) are rendered by one node per size, like the surfel twin does."""
    V = len(settings_list)
    groups = _size_groups(settings_list)
    if len(groups) <= 1:
        out = _RenderViews.apply(means3D, means2D, sh, opacities, scales, rotations, tuple(settings_list), int(flags))
        return list(out[1:1 + V]), out[0], list(out[1 + V:1 + 2 * V]), list(out[1 + 2 * V:1 + 3 * V])
    colors, radii, depths, alphas = [None] * V, [None] * V, [None] * V, [None] * V
    for idx, sets in groups:
        n = len(idx)
        out = _RenderViews.apply(means3D, means2D, sh, opacities, scales, rotations, tuple(sets), int(flags))
        for k, v in enumerate(idx):
            colors[v], radii[v], depths[v], alphas[v] = out[1 + k], out[0][k], out[1 + n + k], out[1 + 2 * n + k]
    return colors, torch.stack(radii), depths, alphas


def topk_absgrad(grad, k, candidates=None, return_indices=False):
    """Device top-k of the densification score ||grad[:, 2:4]||_2 (network.py:876-893; k_num = 12 000,
    configs/base.yaml:30) as the boolean mask the reference builds from torch.topk's indices — radix select in
    libgdr_hip.so (gdr_topk_absgrad), no sort, no score tensor.  candidates: optional bool (N,) restricting the
    selection (the reference's `grad[mask]`); k >= number of candidates selects every candidate.  Returns mask (N,)
    bool, and with return_indices also the selected ids (min(k, candidates),) int64 in no particular order."""
    lib = L.load()
    _require_hip(grad, "grad")
    dev = grad.device
    g = _f32(grad, dev)
    N = int(g.shape[0])
    if g.dim() != 2 or g.shape[1] != 4:
        raise RuntimeError("topk_absgrad: grad must be (N,4)")
    cand = None if candidates is None else candidates.to(device=dev, dtype=torch.uint8).contiguous()
    mask = torch.empty(N, dtype=torch.uint8, device=dev)
    k = max(0, int(k))
    idx = torch.full((min(k, N),), -1, dtype=torch.int32, device=dev) if return_indices else None
    ws = torch.empty(int(lib.gdr_topk_workspace_bytes()), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        L.check(lib.gdr_topk_absgrad(N, _ptr(g), _ptr(cand), k, _ptr(ws), _ptr(mask), _ptr(idx), _stream()),
                "gdr_topk_absgrad")
    mask = mask.bool()
    if return_indices:
        idx = idx[idx >= 0].long()     # fewer than k candidates: the tail stays unwritten
        return mask, idx
    return mask


def screenspace_absgrad_raw(means3D, sh, opacities, scales, rotations, settings_list, gt_images, flags=RAW_ALL, topk=0):
    """SURVEY §8f-2.  What the densification step of the reference computes with
    `vjp(fn, screenspace_point)` (network.py:843-878): loss = mean over all views and pixels of
    (clamp(image, 0, 1) - gt)^2 and its gradient w.r.t. the shared (N,4) means2D carrier
    (columns 0-1 signed, 2-3 sum of |per-pixel terms|) — nothing else.  K6 runs in its loss-only form
    (gdr_composite_forward_lossgrad): it accumulates the MSE and writes d loss / d colour per pixel, never an image;
    the mean2D-only K7 accumulates over the views into ONE (N,4) buffer: no colour / depth / alpha tensors, no
    gradient records, no K8/K9.  gt_images: (V,3,H,W).  Returns (loss, grad (N,4)), and with topk > 0 also
    the indices of the topk largest ||grad[:, 2:4]||_2 (the selection of network.py:878-893, configs/base.yaml:30) from
    the device radix select (topk_absgrad)."""
    lib = L.load()
    with torch.no_grad():
        dev = means3D.device
        N, V = int(means3D.shape[0]), len(settings_list)
        dummy = torch.empty(0, 4, device=dev)
        gt = _f32(gt_images, dev)
        targets = [gt[v] for v in range(V)]
        losses = torch.zeros(V, dtype=torch.float32, device=dev)
        dcolors, radii, _, _, states, keep, _ = _forward_views_impl(
            means3D, dummy, sh, opacities, scales, rotations, tuple(settings_list), int(flags),
            loss_spec=("lossgrad", targets, 1.0 / V, losses))
        loss = losses.mean()   # views share one image size: the mean of per-view means is the global mean
        grad = torch.zeros(N, 4, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            keep2: list = []
            structs = [_settings_struct(settings_list[v], dev, keep2) for v in range(V)]
            mode = k7_views_mode(states[0].H, states[0].W, N) if V > 1 else 0
            sides = _SideViews(dev, 1 if mode else V, states[0].H, states[0].W)  # the views accumulate into `grad` with atomics: any interleaving is valid
            for lo in range(0, V if mode else 0, L.GDR_MAX_VIEWS):     # one launch per <= 8 views on the caller's stream
                n = min(L.GDR_MAX_VIEWS, V - lo)
                s_arr = (L.GdrSettings * n)(*structs[lo:lo + n])
                g_arr, b_arr, i_arr = _view_arrays(states, lo, n, 0)
                dc_ptrs = (C.c_void_p * n)(*[dcolors[lo + k].data_ptr() for k in range(n)])
                L.check(lib.gdr_render_backward_mean2d_views(n, s_arr, N, g_arr, b_arr, i_arr, dc_ptrs, _ptr(grad), int(mode == 1),
                                                             _stream()), "gdr_render_backward_mean2d_views")
            for v, st in enumerate(states if not mode else ()):
                g = st.geom
                g.cov3D = states[0].geom.cov3D
                L.check(lib.gdr_render_backward_mean2d(C.byref(structs[v]), N, C.byref(g), C.byref(st.bin), C.byref(st.img),
                                                       dcolors[v].data_ptr(), _ptr(grad), sides.stream(v)),
                        "gdr_render_backward_mean2d")
            sides.join()
        if topk:
            _, idx = topk_absgrad(grad, int(topk), return_indices=True)
            return loss, grad, idx
    return loss, grad


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                        cov3Ds_precomp, raster_settings):
    """The reference boundary.  Calls that are provably handed the same Gaussians as earlier ones (the per-view loops of
    network.py:827-838, 848-856, 964-972) join a render group: one preprocess-backward for all of them (viewgroup.py);
    everything else is one independent autograd node per call."""
    if not torch.is_grad_enabled():     # evaluation (evaluation.py:169-193, tools/meshExtractor.py:73-106): no autograd node at all
        return forward_raw(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings)[:4]
    viewgroup.note_forward()
    if viewgroup.eligible(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp):
        out = viewgroup.grouped_call(viewgroup.PATH_3D, means3D, means2D, sh, opacities, scales, rotations, raster_settings)
        if out is not None:
            return out
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class _LazyModule(nn.Module):
    """An nn.Module whose module machinery is set up on first use.  The reference builds a NEW `GaussianRasterizer` for
    every render call (lightning/renderer.py:106-126) and uses it once; `nn.Module.__init__` (a dozen ordered dicts) and
    `nn.Module.__call__` (hook dispatch) cost ~20 us of host time per call on a path that is host-bound at the reference's
    own scene sizes.  The object IS an nn.Module (isinstance, repr, .to(), state_dict(), hooks ...): whatever touches the
    machinery initialises it; a plain call of a never-touched instance goes straight to forward (no hook can exist yet)."""

    def __init__(self, raster_settings):
        object.__setattr__(self, "raster_settings", raster_settings)      # (nn.Module.__init__ deferred: see class doc)

    def _module_init(self):
        if "_parameters" not in self.__dict__:
            rs = self.__dict__.get("raster_settings")
            nn.Module.__init__(self)
            object.__setattr__(self, "raster_settings", rs)

    def __getattr__(self, name):
        if "_parameters" not in self.__dict__:      # first touch of the module machinery (nn.Module keeps it in __dict__)
            self._module_init()
            return getattr(self, name)
        return nn.Module.__getattr__(self, name)

    def __setattr__(self, name, value):
        if name != "raster_settings":
            self._module_init()
        nn.Module.__setattr__(self, name, value)

    def __call__(self, *args, **kwargs):
        if "_parameters" not in self.__dict__:
            return self.forward(*args, **kwargs)
        return nn.Module.__call__(self, *args, **kwargs)

    def __setstate__(self, state):
        self._module_init()
        nn.Module.__setstate__(self, state)


class GaussianRasterizer(_LazyModule):

    def markVisible(self, positions):
        lib = L.load()
        _require_hip(positions, "positions")
        rs = self.raster_settings
        with torch.no_grad(), torch.cuda.device(positions.device):
            p = _f32(positions, positions.device)
            view = _f32(rs.viewmatrix, p.device)
            proj = _f32(rs.projmatrix, p.device)
            out = torch.empty(p.shape[0], dtype=torch.uint8, device=p.device)
            L.check(lib.gdr_mark_visible(int(p.shape[0]), _ptr(p), view.data_ptr(), proj.data_ptr(),
                                         _ptr(out), _stream()), "gdr_mark_visible")
        return out.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        e = empty_f32(means3D.device)
        return rasterize_gaussians(
            means3D, means2D, e if shs is None else shs, e if colors_precomp is None else colors_precomp,
            opacities, e if scales is None else scales, e if rotations is None else rotations,
            e if cov3D_precomp is None else cov3D_precomp, self.raster_settings)
