"""View-sharded rendering of one Gaussian set.

The reference renders target views in a serial Python loop per sample
(/root/reference/lightning/network.py:826-838, 848-856, 964-972) and shards only by
scene (DDP, train_lightning.py:71-75).  Views are independent given the Gaussians, so
here rank g of G renders views [g*V/G, (g+1)*V/G): no data-path collective; the only
exchange is an all-gather of the V per-view losses (V floats over RCCL/xGMI; gloo on
CPU in tests) and, for a real training step on shared Gaussians, a sum of the packed
attribute gradients (`allreduce_gaussian_grads`).
"""
from __future__ import annotations

from typing import Callable, Sequence

import os

import torch
import torch.distributed as dist


def shard_views(n_views: int, rank: int, world: int) -> range:
    """Contiguous block partition; the first (n_views % world) ranks get one extra view."""
    base, extra = divmod(n_views, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def render_views(renderer, cams: Sequence, bg_colors, gaussians: dict, device, prex: str = "",
                 screenspace_points=None, stacked: bool = False, raw: bool = False):
    """One `render_img` per camera (same call the reference loop makes); bg_colors may be
    None (keep the renderer's), one tensor, or one per view (network.py:829-830).
    raw=True (fused renderer only): the rasterizer's own layout, see Renderer.render_views."""
    if raw and not (hasattr(renderer, "render_views") and getattr(renderer, "fused", False)):
        raise RuntimeError("raw=True needs the fused HIP renderer")
    if hasattr(renderer, "render_views") and getattr(renderer, "fused", False) and gaussians["centers"].is_cuda:
        return renderer.render_views(cams, bg_colors, gaussians["centers"], gaussians["shs"], gaussians["opacity"],
                                     gaussians["scales"], gaussians["rotations"], device, prex=prex,
                                     screenspace_points=screenspace_points, stacked=stacked, raw=raw)
    outs = []
    for j, cam in enumerate(cams):
        if bg_colors is not None:
            renderer.set_bg_color(bg_colors[j] if isinstance(bg_colors, (list, tuple)) else bg_colors)
        outs.append(renderer.render_img(cam, None, gaussians["centers"], gaussians["shs"], gaussians["opacity"],
                                        gaussians["scales"], gaussians["rotations"], device, prex=prex,
                                        screenspace_points=screenspace_points))
    if stacked:  # same dict of view-stacked tensors the fused path returns (network.py:840)
        return {k: torch.stack([o[k] for o in outs]) for k in outs[0]}
    return outs


def _no_peers() -> bool:
    """True when there is nothing to exchange.  GDR_FORCE_COLLECTIVES=1 keeps the collectives in a
    one-rank group (used to exercise the RCCL calls of the N>1 path on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size() == 1 and os.environ.get("GDR_FORCE_COLLECTIVES", "0") != "1"


class _PendingLosses:
    """An all-gather of per-view losses in flight (gather_view_losses(async_op=True)): wait() returns them in global view
    order — call it where the losses are read (logging), not where they are produced."""

    def __init__(self, work, out, m, sizes):
        self.work, self.out, self.m, self.sizes = work, out, m, sizes

    def wait(self) -> torch.Tensor:
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.sizes is None:
            return self.out
        return torch.cat([self.out[r * self.m: r * self.m + n] for r, n in enumerate(self.sizes)])


def gather_view_losses(local_losses: torch.Tensor, n_views: int | None = None, async_op: bool = False):
    """All-gather of per-view scalar losses in global view order.  Uneven shards are padded
    to the largest shard with NaN and stripped again.  async_op=True (needs n_views): the collective runs on the process
    group's stream and a handle comes back; handle.wait() returns the losses (a step that only logs them joins there).
    A step that ALSO sums gradients over the ranks needs no collective of its own for the losses: they ride in the tail of
    the packed gradient buffer (allreduce_gaussian_grads(view_losses=...))."""
    if _no_peers():
        return _PendingLosses(None, local_losses, 0, None) if async_op else local_losses
    world = dist.get_world_size()
    if async_op and n_views is None:
        raise ValueError("gather_view_losses(async_op=True) needs n_views (the shard sizes must be known without a collective)")
    n_local = torch.tensor([local_losses.numel()], device=local_losses.device, dtype=torch.int64)
    if n_views is None:
        sizes = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(sizes, n_local)
        sizes = [int(s) for s in sizes]
    else:
        sizes = [len(shard_views(n_views, r, world)) for r in range(world)]
    m = max(sizes)
    pad = torch.full((m,), float("nan"), device=local_losses.device, dtype=local_losses.dtype)
    pad[: local_losses.numel()] = local_losses
    out = torch.empty(world * m, device=local_losses.device, dtype=local_losses.dtype)
    if async_op:
        return _PendingLosses(dist.all_gather_into_tensor(out, pad, async_op=True), out, m, sizes)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * m: r * m + sizes[r]] for r in range(world)])


class _Pack:
    """Persistent packed gradient buffer of ONE set of parameter tensors of one dtype (identified by the tensors
    themselves, not by their shapes: two Gaussian sets of equal N — coarse / fine, two models — must never share one)."""
    __slots__ = ("flat", "views", "shard", "refs", "tail")


_PACKS: dict = {}   # (world, ids of the parameters) -> _Pack; the weak references guard against id() reuse


def _grad_pack(params, world, tail: int = 0):
    """tail: extra elements behind the gradients (the per-view losses of a step ride there: allreduce_gaussian_grads)."""
    import weakref
    key = (world, tuple(id(p) for p in params), int(tail))
    pack = _PACKS.get(key)
    if pack is not None and all(r() is p for r, p in zip(pack.refs, params)):
        _PACKS[key] = _PACKS.pop(key)      # most recently used last: the eviction below drops the LEAST recently used set
        return pack
    for k in [k for k, v in _PACKS.items() if any(r() is None for r in v.refs)]:    # sets whose tensors are gone
        _drop_pack(_PACKS.pop(k))
    n = sum(p.numel() for p in params)
    padded = (n + int(tail) + world - 1) // world * world
    pack = _Pack()
    pack.flat = torch.zeros(padded, device=params[0].device, dtype=params[0].dtype)
    pack.views, off = [], 0
    for p in params:
        pack.views.append(pack.flat[off: off + p.numel()].view(p.shape))
        off += p.numel()
    pack.tail = pack.flat[n: n + int(tail)]
    pack.shard = torch.empty(padded // world, device=pack.flat.device, dtype=pack.flat.dtype)
    pack.refs = [weakref.ref(p) for p in params]
    while len(_PACKS) >= 16:     # (an evicted set's .grad tensors keep their storage alive; its next call builds a new buffer
        _drop_pack(_PACKS.pop(next(iter(_PACKS))))   #  and copies once — LRU, so that only sets not reduced for 16 other sets pay that)
    _PACKS[key] = pack
    return pack


def _drop_pack(pack):
    """A packed buffer leaves the cache: its slices must not stay registered as gradient sinks (the registry would keep the
    whole buffer — 472 MB at 2 M Gaussians — alive until the next registration; round-5 advisor finding)."""
    try:
        from . import rasterizer as R
        R.unregister_grad_sinks(pack.views)
    except Exception:      # noqa: BLE001 — (the rasterizer module needs the HIP library; the CPU tests of the collectives do not)
        pass


class _PendingReduce:
    """Collectives in flight (allreduce_gaussian_grads(async_op=True)).  wait(): the caller's stream waits for them — call it
    before anything reads the gradients (the optimizer) or writes the packed buffer again (the next backward)."""

    def __init__(self, works, losses=None):
        self.works = works
        self._losses = losses

    def wait(self):
        for w in self.works:
            w.wait()
        self.works = []
        return self

    @property
    def losses(self):
        """The per-view losses of all ranks in global view order (allreduce_gaussian_grads(view_losses=...)); joins first."""
        self.wait()
        return self._losses


def prepare_grad_sinks(params: Sequence[torch.Tensor], any_device: bool = False, tail=(0, None)) -> None:
    """Create the packed gradient buffer(s) of `params` now and register every slice as the gradient sink of its tensor
    (rasterizer.register_grad_sink): from the next backward on, the multi-view nodes' K9 writes the gradients of these leaves
    straight into the buffer the collectives move — no copy in allreduce_gaussian_grads.  (Called by
    allreduce_gaussian_grads itself, so the first step pays one copy and the later ones none.)"""
    if not (dist.is_available() and dist.is_initialized()):
        return
    world = dist.get_world_size()
    groups: dict = {}
    for p in params:
        groups.setdefault((p.device, p.dtype), []).append(p)
    for (dev, dtype), group in groups.items():
        if dtype != torch.float32 or (dev.type != "cuda" and not any_device):     # (any_device: the CPU tests of the mechanism)
            continue
        from . import rasterizer as R
        pack = _grad_pack(group, world, tail[0] if tail[1] == (dev, dtype) else 0)   # (tail: the buffer that also carries a step's losses)
        for p, v in zip(group, pack.views):
            if p.is_leaf and p.requires_grad and p.is_contiguous():
                R.register_grad_sink(p, v)


def allreduce_gaussian_grads(params: Sequence[torch.Tensor], async_op: bool = False, view_losses: torch.Tensor | None = None,
                             n_views: int | None = None, use_sinks: bool = True):
    """Sum the per-Gaussian attribute gradients over ranks: one packed buffer
    (236 B/Gaussian at SH degree 3) moved as reduce-scatter + all-gather so that all
    7 xGMI links of a GPU carry 1/G of it each, instead of a per-tensor ring all-reduce.

    Participation is unconditional and the buffer layout is rank-invariant: EVERY tensor of `params` takes part
    with its full size (a missing .grad counts as zeros), so a rank whose shard is empty
    (n_views < world) or that produced gradients for only some tensors issues the same collectives with the
    same sizes as every other rank.  After the call every rank holds the summed gradient in every p.grad.

    The packed buffer is PERSISTENT — one per SET OF TENSORS (keyed on the tensors' identity, weakly referenced; tensors of
    different dtypes get one buffer per dtype) — and the gradients the call leaves behind are views of it: no `torch.cat`
    of 472 MB (2 M Gaussians), no copy back.  A training loop that drops its gradients between steps (`p.grad = None`) pays
    ONE copy_ of the fresh gradients into the buffer (a read and a write of the 472 MB); a loop that keeps them
    (`zero_grad(set_to_none=False)`) pays no copy here but a fill plus autograd's accumulate-in-place (read, read, write)
    instead — measured at 2 M Gaussians, one rank with the collectives forced (`bench.py --force-dist --grad-allreduce
    [--keep-grads]`, profiles/r04_bench_rccl_1rank*.json): 4.28 ms per step dropped, 4.55 ms kept: dropping is the cheaper
    loop, and what `bench.py` does.

    (Not chunked: the six gradient tensors of a Gaussian set come out of ONE kernel launch (K8+K9 writes every output), so
    there is no earlier point at which part of the buffer is final, and collectives queued on one communicator run in order —
    chunks would neither start sooner nor overlap each other.)

    Round 5: (a) no copy at all in the steady state — the slices of the packed buffer are registered as the gradient sinks of
    their tensors (prepare_grad_sinks), the multi-view nodes' K9 writes there directly and `.grad` arrives as a view of the
    buffer; (b) async_op=True returns a handle instead of making the caller's stream wait: the collectives run on the process
    group's stream next to whatever the caller enqueues next (the next sample's forward) — handle.wait() before the optimizer
    reads the gradients or the next backward writes the buffer.

    Round 6 — ONE pair of collectives per step (the reference's DDP issues one bucketed all-reduce per step,
    /root/reference/train_lightning.py:71-76): view_losses = this rank's per-view losses (its shard of the n_views views, in
    order).  They ride in the tail of the packed buffer: every rank writes its losses at its GLOBAL view positions and zeros at
    the others, so the sum over the ranks that reduce-scatter + all-gather form anyway IS the all-gather of the losses — no
    separate loss collective (0.12 ms of launch latency for a handful of floats, profiles/r05_bench_rccl_1rank*.json), no
    size exchange, no NaN padding.  With view_losses the call always returns a handle; handle.losses (joins) is the (n_views,)
    tensor in global view order.

    ALIASING (as DDP's gradient_as_bucket_view): after the call `.grad` of every tensor is a VIEW of the persistent packed
    buffer, and with use_sinks=True (default) the next backward's K9 writes the new gradients into that same memory — a caller
    that keeps a reference to an old gradient after `p.grad = None` (manual accumulation, densification statistics) sees it
    overwritten.  Clone what must survive a step, or pass use_sinks=False (the gradients then arrive in fresh tensors and are
    copied into the buffer by the next call: one read + write of the buffer per step)."""
    with_losses = view_losses is not None
    if with_losses and n_views is None:
        raise ValueError("allreduce_gaussian_grads(view_losses=...) needs n_views")
    if _no_peers():
        if with_losses:
            return _PendingReduce([], view_losses)
        return _PendingReduce([]) if async_op else None
    world = dist.get_world_size()
    params = list(params)
    if not params:
        if with_losses:
            return _PendingReduce([], gather_view_losses(view_losses, n_views))
        return _PendingReduce([]) if async_op else None
    works = []
    gathered = None
    groups: dict = {}
    for p in params:     # one pack per dtype (and device), the order within a group as given
        groups.setdefault((p.device, p.dtype), []).append(p)
    carrier = None       # the group whose buffer carries the losses: the first one of the losses' dtype and device
    if with_losses:
        carrier = next((k for k in groups if k == (view_losses.device, view_losses.dtype)), None)
        if carrier is None:      # (no gradient buffer of that dtype: the losses take their own collective)
            gathered = gather_view_losses(view_losses, n_views)
    for gkey, group in groups.items():
        pack = _grad_pack(group, world, n_views if gkey == carrier else 0)
        if gkey == carrier:
            mine = shard_views(n_views, dist.get_rank(), world)
            if view_losses.numel() != len(mine):
                raise ValueError(f"view_losses has {view_losses.numel()} entries, this rank's shard of {n_views} views has {len(mine)}")
            pack.tail.zero_()
            if len(mine):
                pack.tail[mine.start: mine.stop].copy_(view_losses.detach().reshape(-1))
            gathered = pack.tail
        for p, v in zip(group, pack.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr() or p.grad.dtype != v.dtype:
                v.copy_(p.grad)
        w1 = dist.reduce_scatter_tensor(pack.shard, pack.flat, op=dist.ReduceOp.SUM, async_op=async_op)
        if async_op and dist.get_backend() != "nccl":
            w1.wait()      # (RCCL runs a group's collectives in issue order on its stream; gloo's worker threads do not)
        w2 = dist.all_gather_into_tensor(pack.flat, pack.shard, async_op=async_op)
        works += [w1, w2] if async_op else []
        for p, v in zip(group, pack.views):
            p.grad = v
    if params[0].is_cuda and use_sinks:
        prepare_grad_sinks(params, tail=(n_views if carrier is not None else 0, carrier))
    if with_losses:
        return _PendingReduce(works, gathered)
    return _PendingReduce(works) if async_op else None
