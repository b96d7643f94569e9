"""Render groups — a fast path for the UNCHANGED caller, for both boundaries:
`diff_gaussian_rasterization` (/root/reference/lightning/renderer.py:250-259) and, since round 4,
`diff_surfel_rasterization` (/root/reference/lightning/renderer_2dgs.py:224-234).

The reference renders the views of one Gaussian set one `render_img` at a time
(/root/reference/lightning/network.py:827-838, 848-856, 964-972): every call builds a new settings tuple, applies
sigmoid / exp / normalize again (/root/reference/lightning/renderer.py:225-230), allocates a new (N,4) carrier and calls
`GaussianRasterizer` — and Lightning back-propagates ONCE through all those graphs.  Seen from the rasterizer the V calls
are V unrelated autograd nodes: V preprocess-backward passes (K8+K9, 1.2 GB each at 2 M Gaussians), V full-size gradient
sets that autograd adds up (another 1.4 GB per view), V activation backward chains.

Here consecutive calls that are PROVABLY handed the same Gaussians form a group:

    caller's tensors ──► _Hub (one node per group) ──► aliases ──► _GroupView_0 ──► view 0's images
                                                          ├──────► _GroupView_1 ──► view 1's images
                                                          └──────► ...

* forward: every call is ONE native call (gdr_forward_view / gsr_forward_view: K1 .. K6 exactly as an ungrouped call);
* backward: `_GroupView_j.backward` runs K7 of its view only (its (N,4) means2D gradient is the head of the view's gradient
  records) and returns None for the shared inputs; autograd's dependency counting runs `_Hub.backward` after every view
  node of the pass, and the hub runs ONE multi-view K8+K9 (`g?r_preprocess_backward_views`: inputs read once, the per-view
  gradients summed in registers, every output written once) and hands the sums to the producers of the FIRST call's
  tensors — the later calls' activation chains never run backward.  Which views took part in the pass: every view node
  tags its K7 result with the id of the running backward pass (`torch._C._current_graph_task_id()`), the hub takes the
  ones that carry its own.  Views that ran K7 in an earlier pass the hub was not part of
  (`torch.autograd.functional.vjp` w.r.t. the carrier only, network.py:872) are dropped when the next pass parks its first
  result.

"Provably the same": every input's autograd provenance is hashed (`_signature`): the chain of whitelisted, deterministic
ops (select / sigmoid / exp / the ops of F.normalize ...) with their saved scalars down to leaves (identity + version
counter) or to opaque nodes (identity).  Equal signatures = the same function of the same sources.  What the signature
cannot see — an in-place edit of a non-leaf source under no_grad between two calls — is caught on the device: every later
call's activated tensors are compared with the group's bit for bit (gdr_same_as, next to K1; the verdict travels to the
host with the duplicate count) and THAT CALL is rendered again as an ordinary independent node from its own tensors
(round 4; it used to raise).  Anything not provable (colors_precomp, cov3D_precomp, no_grad, an op outside the whitelist)
takes the ordinary one-node-per-call path.

What a caller can observe: the gradients of the group reach the producers of the FIRST call's tensors;
the activation tensors of LATER calls (their `sigmoid(opacity)` ...) receive none — so a call whose inputs carry tensor
hooks or `retain_grad()` is never grouped.  A caller that back-propagates after every single view gains nothing from
groups and pays their bookkeeping: after two single-view passes in a row grouping pauses until the caller renders several
views per pass again.  The mechanism leans on two private pieces of torch — `torch._C._current_graph_task_id` and the
`_saved_*` attributes of autograd nodes; both are probed at import, and if either is missing every call is an ordinary
node (one warning).  GDR_GROUP_VIEWS=0 switches grouping off.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import warnings
import weakref

import torch

from . import _lib as L

GROUP_VIEWS = os.environ.get("GDR_GROUP_VIEWS", "1") != "0"
MAX_VIEWS_PER_GROUP = 64
_LOCK = threading.RLock()
_GROUPS: dict = {}                # (path name, signature key) -> weakref to _Group (the view nodes hold the group alive)

# deterministic ops whose output is a function of their (tracked) inputs and the listed saved scalars only
_OPS = {
    "SigmoidBackward0": (), "ExpBackward0": (), "DivBackward0": (), "MulBackward0": (), "AddBackward0": ("_saved_alpha",),
    "SelectBackward0": ("_saved_dim", "_saved_index", "_saved_self_sym_sizes"),
    "ExpandBackward0": ("_saved_self_sym_sizes",), "ClampMinBackward0": ("_saved_min",),
    "LinalgVectorNormBackward0": ("_saved_ord", "_saved_dim", "_saved_keepdim"),
    "NormBackward1": ("_saved_p", "_saved_dim", "_saved_keepdim"),
    "ViewBackward0": ("_saved_self_sym_sizes",), "UnsafeViewBackward0": ("_saved_self_sym_sizes",),
    "ReshapeAliasBackward0": ("_saved_self_sym_sizes",), "SqueezeBackward1": ("_saved_dim", "_saved_self_sym_sizes"),
    "UnsqueezeBackward0": ("_saved_dim",), "AliasBackward0": (),
}


def _probe_torch() -> str:
    """'' if the private torch pieces this module leans on behave as expected, else what is missing."""
    if not hasattr(torch._C, "_current_graph_task_id"):
        return "torch._C._current_graph_task_id is missing"
    try:
        x = torch.ones(2, 4, requires_grad=True)
        probes = (torch.sigmoid(x), torch.exp(x), torch.nn.functional.normalize(x), x[0])
        seen = set()
        for t in probes:
            stack = [t.grad_fn]
            while stack:
                fn = stack.pop()
                name = type(fn).__name__
                if name == "AccumulateGrad":
                    if fn.variable is not x:
                        return "AccumulateGrad.variable does not return the leaf"
                    continue
                if name not in _OPS:
                    return f"unknown autograd node {name} behind a reference activation"
                for a in _OPS[name]:
                    getattr(fn, a)
                seen.add(name)
                stack += [n for n, _ in fn.next_functions if n is not None]
        if not {"SigmoidBackward0", "ExpBackward0", "DivBackward0", "SelectBackward0"} <= seen:
            return "the reference activations map to other autograd nodes than expected"
        if torch._C._current_graph_task_id() != -1:
            return "_current_graph_task_id() outside a backward pass is not -1"
    except Exception as exc:      # noqa: BLE001 — anything unexpected means: do not lean on it
        return f"{type(exc).__name__}: {exc}"
    return ""


_PROBLEM = _probe_torch() if GROUP_VIEWS else ""
if _PROBLEM:
    warnings.warn("generativedensification_amd: render groups are off — this torch build does not match what they rely on "
                  f"({_PROBLEM}); every rasterizer call is an independent autograd node (correct, slower backward).")
    GROUP_VIEWS = False

# ---- callers that back-propagate after every single view (no gain, only bookkeeping): pause grouping -------------------
_calls_since_backward = 0
_solo_passes = 0          # consecutive backward passes that were preceded by exactly one forward call


def note_forward():
    """Called by every differentiable forward call of the boundary (grouped or not)."""
    global _calls_since_backward
    _calls_since_backward += 1


def note_backward():
    """Called by every backward entry of the boundary (grouped or not): closes the count of forward calls of this pass."""
    global _calls_since_backward, _solo_passes
    if _calls_since_backward == 0:
        return                      # a later node of the same pass
    _solo_passes = _solo_passes + 1 if _calls_since_backward == 1 else 0
    _calls_since_backward = 0


def _hashable(v):
    if isinstance(v, (list, tuple)):
        return tuple(_hashable(x) for x in v)
    if isinstance(v, torch.Tensor):
        raise TypeError
    try:
        hash(v)
        return v
    except TypeError:
        return repr(v)


def _node_sig(fn, depth, hold):
    name = type(fn).__name__
    if name == "AccumulateGrad":
        v = fn.variable
        hold.append(v)
        return ("leaf", id(v), v._version, v.data_ptr(), tuple(v.shape), tuple(v.stride()))
    attrs = _OPS.get(name)
    nxt = fn.next_functions
    if attrs is None or depth > 8 or any(n is None for n, _ in nxt):
        hold.append(fn)                       # opaque: the node's identity (kept alive so that the id stays unique)
        return ("node", id(fn))
    try:
        saved = tuple(_hashable(getattr(fn, a)) for a in attrs)
    except (AttributeError, TypeError, RuntimeError):
        hold.append(fn)
        return ("node", id(fn))
    return (name, saved, tuple((_node_sig(n, depth + 1, hold), nr) for n, nr in nxt))


def _signature(t: torch.Tensor, hold: list):
    """Hashable provenance of a tensor: equal signatures => equal values (module docstring).  `hold` receives the objects
    whose id() enters the signature; whoever keeps the signature keeps them.  (Dtype conversions — ToCopy — are opaque:
    `.half().float()` and `.bfloat16().float()` of one source are different values with the same saved scalars.)"""
    fn = t.grad_fn
    meta = (tuple(t.shape), t.dtype, t.device.index)
    if fn is None:        # a leaf (or a tensor outside any graph): itself
        hold.append(t)
        return ("tensor", id(t), t._version, t.data_ptr(), tuple(t.stride())) + meta
    return (_node_sig(fn, 0, hold), t.output_nr, t._version) + meta


class _Group:
    __slots__ = ("path", "key", "hold", "orig", "f32", "hub_out", "token", "one", "n_views", "pending", "dev", "N", "M", "lock",
                 "closed", "__weakref__")

    def __init__(self, path, key, hold, orig, dev):
        self.path, self.key, self.hold, self.orig, self.dev = path, key, hold, orig, dev
        self.f32 = None
        self.hub_out = self.token = self.one = None
        self.n_views = 0
        self.pending = {}
        self.lock = threading.RLock()
        self.closed = False       # set by the hub's backward: its graph may be freed, later calls open a new group


def _observed(t: torch.Tensor) -> bool:
    """A non-leaf input somebody watches: in a group the later calls' activation tensors receive no gradient."""
    return t.grad_fn is not None and (t.retains_grad or bool(t._backward_hooks))


def eligible(means3D, sh, colors_precomp, opacities, scales, rotations, precomp) -> bool:
    """Can this call join / open a render group?  (precomp: cov3D_precomp / transMat_precomp)"""
    if not (GROUP_VIEWS and torch.is_grad_enabled() and means3D.is_cuda and sh.numel() and not colors_precomp.numel()
            and scales.numel() and rotations.numel() and not precomp.numel() and means3D.shape[0] > 0):
        return False
    ts = (means3D, sh, opacities, scales, rotations)
    if not any(t.requires_grad for t in ts) or any(_observed(t) for t in ts):
        return False
    return _solo_passes < 2


def _find_group(path, tensors, dev, raster_settings):
    hold: list = []
    rs = raster_settings     # one group = one image size, SH degree and scale modifier: what a multi-view K8+K9 launch shares
    key = (path.name, int(rs.image_height), int(rs.image_width), int(rs.sh_degree), float(rs.scale_modifier)) \
        + tuple(_signature(t, hold) for t in tensors)
    with _LOCK:
        ref = _GROUPS.get(key)
        grp = ref() if ref is not None else None
        if grp is not None and grp.n_views < MAX_VIEWS_PER_GROUP and not grp.closed:
            return grp, False
        grp = _Group(path, key, hold, tuple(tensors), dev)
        _GROUPS[key] = weakref.ref(grp)
        for k in [k for k, r in _GROUPS.items() if r() is None]:     # dead groups: their ids may be reused
            del _GROUPS[k]
        return grp, True


def _same_as_pairs(grp, tensors, R):
    """Later call of a group: the (tensor, group's tensor) pairs whose equality the signature asserts but that are not the
    very same memory — the forward compares them on the device next to K1."""
    pairs = []
    for k, (t, ref) in enumerate(zip(tensors, grp.f32)):
        if t is grp.orig[k]:
            continue
        t32 = R._f32(t, grp.dev)
        if t32.shape != ref.shape:
            raise R.GroupMismatch("render group: equal provenance but different shapes")
        if t32.data_ptr() != ref.data_ptr():
            pairs.append((t32, ref))
    return pairs


# ---- what differs between the two boundaries ---------------------------------------------------------------------------
class _Path3D:
    name, floats, scale_cols, n_out = "3dgs", 16, 3, 4

    @staticmethod
    def supports(sh, raster_settings):
        return True

    @staticmethod
    def forward(grp, raster_settings, same_as):
        from . import rasterizer as R
        e = R.empty_f32(grp.dev)
        means3D, sh, opacities, scales, rotations = grp.f32
        color, radii, depth, alpha, st, keep = R.forward_raw(means3D, sh, e, opacities, scales, rotations, e, raster_settings,
                                                             same_as=same_as)
        return (color, radii, depth, alpha), st, keep[7:]

    @staticmethod
    def k7(lib, s, N, st, grads, H, W, dev, recs, keep):
        from . import rasterizer as R
        gc, _, gd, ga = grads
        gc = R._f32(gc, dev) if gc is not None else torch.zeros(3, H, W, dtype=torch.float32, device=dev)
        gd = None if gd is None else R._f32(gd, dev)
        ga = None if ga is None else R._f32(ga, dev)
        keep += [gc, gd, ga]
        gin = L.GdrGradInputs(gc.data_ptr(), R._ptr(gd), R._ptr(ga))
        L.check(lib.gdr_render_backward(C.byref(s), N, C.byref(st.geom), C.byref(st.bin), C.byref(st.img), C.byref(gin),
                                        recs.data_ptr(), R._stream()), "gdr_render_backward")

    @staticmethod
    def view_means2d(lib, s, N, st, radii, recs, dev):
        """(N,4) means2D gradient of this view: K7 accumulates it straight into the head of the 64-byte record."""
        return recs.view(N, 16)[:, :4]

    @staticmethod
    def inputs(N, M, f32, flags=0):
        from . import rasterizer as R
        means3D, sh, opacities, scales, rotations = f32
        e = R.empty_f32(means3D.device)
        return R._inputs_struct(N, M, means3D, opacities, sh, e, scales, rotations, e, flags)

    @staticmethod
    def k9_views(lib, n, s_arr, inp, g_arr, r_arr, rec_arr, g, accumulate, stream):
        from . import rasterizer as R
        gout = L.GdrGradOutputs(R._ptr(g["means3D"]), R._ptr(g["means2D"]), R._ptr(g["shs"]), None, R._ptr(g["opacities"]),
                                R._ptr(g["scales"]), R._ptr(g["rotations"]), None, None, accumulate, 0)
        L.check(lib.gdr_preprocess_backward_views(n, s_arr, C.byref(inp), g_arr, r_arr, rec_arr, C.byref(gout), stream),
                "gdr_preprocess_backward_views")

    @staticmethod
    def share_geom(g_arr, first_state):
        for k in range(len(g_arr)):
            g_arr[k].cov3D = first_state.geom.cov3D      # view-independent: any view's copy


class _PathSurfel:
    name, floats, scale_cols, n_out = "surfel", L.GSR_GRAD_FLOATS, 2, 3

    @staticmethod
    def supports(sh, raster_settings):
        nb = (int(raster_settings.sh_degree) + 1) ** 2    # gsr_preprocess_backward_views (include/gsr.h): at SH degree 1 / 3
        return not ((3 * nb) % 4 == 0 and int(sh.shape[1]) != nb)     # the SH block must be exactly (deg+1)^2 rows

    @staticmethod
    def forward(grp, raster_settings, same_as):
        from . import rasterizer as R, surfel_rasterizer as S
        e = R.empty_f32(grp.dev)
        means3D, sh, opacities, scales, rotations = grp.f32
        color, radii, allmap, st, keep = S.forward_raw(means3D, sh, e, opacities, scales, rotations, e, raster_settings,
                                                       same_as=same_as)
        return (color, radii, allmap), st, keep[8:]

    @staticmethod
    def k7(lib, s, N, st, grads, H, W, dev, recs, keep):
        from . import rasterizer as R
        gc, _, gm = grads
        gc = R._f32(gc, dev) if gc is not None else torch.zeros(3, H, W, dtype=torch.float32, device=dev)
        gm = None if gm is None else R._f32(gm, dev)
        keep += [gc, gm]
        gin = L.GsrGradInputs(gc.data_ptr(), R._ptr(gm))
        L.check(lib.gsr_render_backward(C.byref(s), N, C.byref(st.geom), C.byref(st.bin), C.byref(st.img), C.byref(gin),
                                        recs.data_ptr(), R._stream()), "gsr_render_backward")

    @staticmethod
    def view_means2d(lib, s, N, st, radii, recs, dev):
        """(N,4) means2D gradient of this view: the densification signal K9s forms from the record (dL/dT words x depth x
        W/2 | H/2 and its |.| twin) — gsr_means2d_of_view, one small launch."""
        from . import rasterizer as R
        out = torch.empty(N, 4, dtype=torch.float32, device=dev)
        L.check(lib.gsr_means2d_of_view(C.byref(s), N, C.byref(st.geom), radii.data_ptr(), recs.data_ptr(), out.data_ptr(),
                                        R._stream()), "gsr_means2d_of_view")
        return out

    @staticmethod
    def inputs(N, M, f32, flags=0):
        from . import rasterizer as R, surfel_rasterizer as S
        means3D, sh, opacities, scales, rotations = f32
        e = R.empty_f32(means3D.device)
        return S._inputs_struct(N, M, means3D, opacities, sh, e, scales, rotations, e, flags)

    @staticmethod
    def k9_views(lib, n, s_arr, inp, g_arr, r_arr, rec_arr, g, accumulate, stream):
        from . import rasterizer as R
        gout = L.GsrGradOutputs(R._ptr(g["means3D"]), R._ptr(g["means2D"]), R._ptr(g["shs"]), None, R._ptr(g["opacities"]),
                                R._ptr(g["scales"]), R._ptr(g["rotations"]), None, None, accumulate, 0)
        L.check(lib.gsr_preprocess_backward_views(n, s_arr, C.byref(inp), g_arr, r_arr, rec_arr, C.byref(gout), stream),
                "gsr_preprocess_backward_views")

    @staticmethod
    def share_geom(g_arr, first_state):
        pass


PATH_3D, PATH_SURFEL = _Path3D, _PathSurfel


class _Hub(torch.autograd.Function):
    """One per group: hands aliases of the caller's tensors to the view nodes; its backward is the group's K8+K9."""

    @staticmethod
    def forward(ctx, grp, means3D, sh, opacities, scales, rotations):
        ctx.grp_ref = weakref.ref(grp)
        ctx.set_materialize_grads(False)
        ctx.in_dtypes = tuple(t.dtype for t in (means3D, sh, opacities, scales, rotations))
        ctx.versions = tuple(t._version for t in (means3D, sh, opacities, scales, rotations))
        # `token`: a one-element device tensor the view nodes return a (meaningless) gradient for — an autograd node whose
        # incoming gradients are all undefined is queued to the CPU worker, i.e. handed to the thread that called
        # backward() and back: two thread hops per pass, ~300 us measured per single-view pass
        token = torch.zeros(1, dtype=torch.float32, device=means3D.device)
        return (means3D.view_as(means3D), sh.view_as(sh), opacities.view_as(opacities), scales.view_as(scales),
                rotations.view_as(rotations), token)

    @staticmethod
    def backward(ctx, g_m, g_s, g_o, g_sc, g_r, g_token):
        from . import rasterizer as R
        grp = ctx.grp_ref()
        if grp is None:
            return (None,) * 6
        task = torch._C._current_graph_task_id()
        grp.closed = True
        with grp.lock:
            views = [grp.pending[j] for j in sorted(grp.pending) if grp.pending[j]["task"] == task]
            grp.pending.clear()           # (incl. K7 results of passes this hub was not part of)
        if not views:
            return (None,) * 6
        for t, v in zip(grp.orig, ctx.versions):
            if t._version != v:
                raise RuntimeError("one of the variables needed for gradient computation has been modified by an inplace "
                                   "operation (render group inputs)")
        lib, path = L.load(), grp.path
        dev, N, M = grp.dev, grp.N, grp.M
        f32 = dict(dtype=torch.float32, device=dev)
        g = dict(means3D=torch.empty(N, 3, **f32), means2D=torch.empty(N, 4, **f32), shs=torch.empty(N, M, 3, **f32),
                 opacities=torch.empty(N, 1, **f32), scales=torch.empty(N, path.scale_cols, **f32),
                 rotations=torch.empty(N, 4, **f32))
        with torch.cuda.device(dev):
            keep: list = []
            inp = path.inputs(N, M, grp.f32)
            for lo in range(0, len(views), L.GDR_MAX_VIEWS):
                grp_v = views[lo:lo + L.GDR_MAX_VIEWS]
                n = len(grp_v)
                s_arr = (L.GdrSettings * n)(*[v["s"] for v in grp_v])
                g_arr = (L.GdrGeom * n)(*[v["state"].geom for v in grp_v])
                path.share_geom(g_arr, grp_v[0]["state"])
                r_arr = (C.c_void_p * n)(*[v["radii"].data_ptr() for v in grp_v])
                rec_arr = (C.c_void_p * n)(*[v["recs"].data_ptr() for v in grp_v])
                path.k9_views(lib, n, s_arr, inp, g_arr, r_arr, rec_arr, g, 1 if lo > 0 else 0, R._stream())
                keep.append(grp_v)
        grads = [g["means3D"], g["shs"], g["opacities"], g["scales"], g["rotations"]]
        grads = [t.reshape(o.shape) if t.dtype == dt else t.reshape(o.shape).to(dt)
                 for t, o, dt in zip(grads, grp.orig, ctx.in_dtypes)]
        return (None, *grads)


class _GroupView(torch.autograd.Function):
    """One per call: ONE native forward call (K1 .. K6 of the view); K7 backward."""

    @staticmethod
    def forward(ctx, grp, j, raster_settings, same_as, means2D, token, m, s, o, sc, r):
        outs, st, keep_rest = grp.path.forward(grp, raster_settings, same_as)
        ctx.grp, ctx.j, ctx.raster_settings, ctx.state, ctx.radii = grp, j, raster_settings, st, outs[1]
        ctx.keep_rest = list(keep_rest)
        ctx.means2D_shape, ctx.means2D_dtype = tuple(means2D.shape), means2D.dtype
        ctx.mark_non_differentiable(outs[1])
        return outs

    @staticmethod
    def backward(ctx, *grads):
        lib = L.load()
        note_backward()
        grp, st = ctx.grp, ctx.state
        dev, N, path = grp.dev, grp.N, grp.path
        with torch.cuda.device(dev):
            keep: list = []
            s = ctx.keep_rest[-1]       # the settings struct of the forward (its device tensors are in keep_rest too)
            recs = torch.empty(N * path.floats, dtype=torch.float32, device=dev)   # one gradient record per Gaussian
            st.bin.grad_rec_cleared = 0
            path.k7(lib, s, N, st, grads, st.H, st.W, dev, recs, keep)
            head = path.view_means2d(lib, s, N, st, ctx.radii, recs, dev)
        cols = ctx.means2D_shape[1] if len(ctx.means2D_shape) == 2 else 4
        if cols == 3:     # legacy caller (point_decoder/layers/gaussian_renderer.py): xy signed, z = 0
            gm2 = torch.cat([head[:, :2], torch.zeros_like(head[:, :1])], dim=1)
        else:
            gm2 = head[:, :min(cols, 4)].contiguous()
        if gm2.dtype != ctx.means2D_dtype:
            gm2 = gm2.to(ctx.means2D_dtype)
        task = torch._C._current_graph_task_id()
        with grp.lock:
            for k in [k for k, e in grp.pending.items() if e["task"] != task]:
                del grp.pending[k]      # K7 results of an earlier pass no hub collected (a vjp w.r.t. the carrier): free them
            grp.pending[ctx.j] = dict(recs=recs, state=st, radii=ctx.radii, s=s, keep=(keep, ctx.keep_rest), task=task)
        return (None, None, None, None, gm2, grp.one, None, None, None, None, None)


def grouped_call(path, means3D, means2D, sh, opacities, scales, rotations, raster_settings):
    """rasterize_gaussians of `path` for an eligible call (see `eligible`): the call joins / opens its render group.
    Returns the boundary's outputs, or None if the call must be rendered as an ordinary node (its values differ from the
    group's although its provenance matches: the caller falls back)."""
    from . import rasterizer as R
    R._require_hip(means3D, "means3D")
    dev = means3D.device
    tensors = (means3D, sh, opacities, scales, rotations)
    grp, new = _find_group(path, tensors, dev, raster_settings)
    same_as = None
    try:
        with grp.lock:
            if new:
                grp.f32 = tuple(R._f32(t, dev) for t in tensors)
                grp.N, grp.M = int(means3D.shape[0]), int(sh.shape[1])
                if grp.f32[2].numel() != grp.N:
                    raise RuntimeError("opacities must have N elements")
                *grp.hub_out, grp.token = _Hub.apply(grp, *tensors)
                grp.one = torch.ones(1, dtype=torch.float32, device=dev)
            else:
                same_as = _same_as_pairs(grp, tensors, R)
            j = grp.n_views
            grp.n_views += 1
        return _GroupView.apply(grp, j, raster_settings, same_as, means2D, grp.token, *grp.hub_out)
    except R.GroupMismatch:
        # equal provenance, different values (a source edited in place outside autograd's view): this call is an ordinary
        # node on its own tensors, as the reference's would be; the group takes no further calls
        grp.closed = True
        return None
