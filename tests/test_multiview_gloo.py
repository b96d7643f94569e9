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
        q.put((rank, allv.tolist(), first[0], first[1]))
    finally:
        dist.destroy_process_group()


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
