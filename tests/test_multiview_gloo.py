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
        losses = torch.stack([o["image"].reshape(()) for o in outs])
        losses.sum().backward()
        allv = gather_view_losses(losses.detach(), n_views)
        allreduce_gaussian_grads(list(g.values()))
        q.put((rank, allv.tolist(), g["centers"].grad[0, 0].item(), g["rotations"].grad is None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_views", [4, 5])
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
    for rank, allv, gsum, rot_none in res:
        assert allv == expect, (rank, allv)          # global view order, both ranks hold all V losses
        assert gsum == float(sum(range(n_views)))    # d/dcenters summed over ALL views after the reduce
        assert rot_none                              # tensors without grad are skipped


def test_single_process_is_a_no_op():
    x = torch.arange(3.0)
    assert gather_view_losses(x) is x
    allreduce_gaussian_grads([torch.ones(2, requires_grad=True)])
