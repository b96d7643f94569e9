"""CPU: the host logic of the render groups (generativedensification_amd/viewgroup.py) that needs no GPU — the provenance
signature, the import-time probe of the private torch pieces it leans on, the rules that keep a call out of a group, and
the single-view-pass pause.  (The grouped kernels themselves: tests/test_gpu_viewgroup.py.)"""
import torch

from generativedensification_amd import viewgroup as G


def _acts(leaves, i=0):
    return (leaves["centers"][i], leaves["shs"][i], torch.sigmoid(leaves["opacity"][i]), torch.exp(leaves["scales"][i]),
            torch.nn.functional.normalize(leaves["rotations"][i]))


def _leaves(n=7, B=2):
    g = torch.Generator().manual_seed(0)
    shapes = dict(centers=(B, n, 3), shs=(B, n, 4, 3), opacity=(B, n, 1), scales=(B, n, 3), rotations=(B, n, 4))
    return {k: torch.randn(*s, generator=g).requires_grad_(True) for k, s in shapes.items()}


def test_probe_accepts_this_torch_and_names_what_is_missing():
    assert G._probe_torch() == "" and G._PROBLEM == ""
    saved = dict(G._OPS)
    try:
        del G._OPS["SigmoidBackward0"]          # a torch whose sigmoid maps to another autograd node
        assert "SigmoidBackward0" in G._probe_torch()
    finally:
        G._OPS.clear()
        G._OPS.update(saved)
    real = torch._C._current_graph_task_id
    try:
        del torch._C._current_graph_task_id
        assert "_current_graph_task_id" in G._probe_torch()
    finally:
        torch._C._current_graph_task_id = real


def test_signature_equal_for_the_reference_loop_and_different_for_everything_else():
    """The per-view loop of network.py:827-838 re-activates the same leaves every call: equal signatures.  Another sample,
    another op, an in-place edit of a leaf (version counter), a dtype round trip: different."""
    lv = _leaves()

    def sig(ts):
        hold = []
        return tuple(G._signature(t, hold) for t in ts), hold
    a, hold_a = sig(_acts(lv))
    b, _ = sig(_acts(lv))
    assert a == b
    assert sig(_acts(lv, 1))[0] != a                                        # another sample of the batch
    other = list(_acts(lv))
    other[2] = torch.sigmoid(lv["opacity"][0] * 1.0)                         # one more op in the chain
    assert sig(other)[0] != a
    other[2] = torch.sigmoid(lv["opacity"][0]).half().float()                # dtype round trips are opaque nodes
    c1, hold_c = sig(other)
    other[2] = torch.sigmoid(lv["opacity"][0]).bfloat16().float()
    assert sig(other)[0] != c1 and c1 != a
    with torch.no_grad():
        lv["scales"].add_(1.0)                                               # a leaf edited in place: its version enters
    assert sig(_acts(lv))[0] != a
    del hold_a, hold_c


def test_eligibility_rules():
    lv = _leaves()
    e = torch.empty(0)

    def elig(ts, colors=e, precomp=e):
        m, s, o, sc, r = ts
        return G.eligible(m, s, colors, o, sc, r, precomp)
    saved = G.GROUP_VIEWS
    G.GROUP_VIEWS, G._solo_passes = True, 0
    try:
        ts = _acts(lv)
        assert not elig(ts)                                   # CPU tensors: never (there is no CPU path at all)
        fake = [t.detach().requires_grad_(t.requires_grad) for t in ts]
        # the device test aside, walk the other rules on a stand-in whose `is_cuda` says yes
        class Cuda(torch.Tensor):
            is_cuda = True
        cu = [t.as_subclass(Cuda) for t in ts]
        assert elig(cu)
        assert not elig(cu, colors=torch.ones(7, 3)) and not elig(cu, precomp=torch.ones(7, 6))
        with torch.no_grad():
            assert not elig(cu)
        watched = list(cu)
        w = torch.sigmoid(lv["opacity"][0])
        w.retain_grad()
        watched[2] = w.as_subclass(Cuda)
        watched[2].retain_grad()
        assert not elig(watched)                              # a watched activation tensor: ordinary node
        hooked = list(cu)
        h = torch.exp(lv["scales"][0])
        h.register_hook(lambda g: g)
        hooked[3] = h
        assert G._observed(h) and not G._observed(lv["scales"])      # hooks on LEAVES are fine (they get the group's sums)
        G._solo_passes = 2
        assert not elig(cu)                                   # paused after two single-view passes
        del fake
    finally:
        G.GROUP_VIEWS, G._solo_passes = saved, 0


def test_single_view_passes_pause_and_resume():
    G._calls_since_backward, G._solo_passes = 0, 0
    for _ in range(2):
        G.note_forward()
        G.note_backward()
        G.note_backward()            # a second node of the same pass: no effect
    assert G._solo_passes == 2
    for _ in range(4):               # several views, then one pass
        G.note_forward()
    G.note_backward()
    assert G._solo_passes == 0 and G._calls_since_backward == 0
