"""The graphed-callable route of ``Architecture`` (glam_amd.graphs.GraphedCallable): the reference's eager training loop
(src_1gp/trainer.py:286-304 — zero_grad, ``model(batch)`` on a FRESH device copy of the batch, ``loss.backward()``, ``optimizer.step()``)
unchanged, replayed from hipGraphs once a batch content has been seen before."""
import copy

import numpy as np
import pytest
import torch

from glam_amd import graphs, model
from glam_amd.data import Batch, DataLoader, synth_batch, synth_molecule

pytestmark = pytest.mark.gpu


def _net(device, **kw):
    cfg = dict(mol_block="_TripletMessage", message_steps=2, mol_readout="GlobalPool5", e_dim=64, graph_norm="_None", graph_do="_None()",
               end_do="_None()", pre_act="ReLU", graph_act="ReLU", flat_act="ReLU")
    cfg.update(kw)
    return model.Architecture(**cfg).to(device)


def _fresh(b):
    """What the reference's loop hands the model: a new object with new device tensors of the same content (trainer.py:294)."""
    out = Batch(x=b.x.clone(), edge_index=b.edge_index.clone(), edge_attr=b.edge_attr.clone(), y=b.y.clone(), batch=b.batch.clone())
    out.num_graphs = b.num_graphs
    return out


def _loss(out, b):
    return torch.nn.functional.mse_loss(out.view(-1), b.y.view(-1))


@pytest.mark.parametrize("optimizer", ["torch", "glam"])
@pytest.mark.parametrize("fresh", [True, False])
def test_unchanged_training_loop_follows_the_eager_trajectory(device, optimizer, fresh):
    """Five epochs of the reference's loop over three batches (below 512 atoms each: the captured and the eager step launch the same
    kernels): route on vs ``graphed_call = False`` — the same losses and parameters bit for bit, with graphs actually replayed."""
    from glam_amd import optim
    rng = np.random.default_rng(21)
    mols = [synth_molecule(rng) for _ in range(24)]
    torch.manual_seed(7)
    net0 = _net(device)
    results = []
    for routed in (False, True):      # (visits 1 and 2 of a content are eager, the third captures, the fourth replays)
        net = copy.deepcopy(net0)
        net.graphed_call = routed
        opt = torch.optim.Adam(net.parameters(), lr=1e-3) if optimizer == "torch" else optim.Adam(net.parameters(), lr=1e-3)
        loader = DataLoader(mols, batch_size=8, device=device)
        losses = []
        for _epoch in range(5):
            for b in loader:
                data = _fresh(b) if fresh else b
                opt.zero_grad()
                loss = _loss(net(data), data)
                loss.backward()
                opt.step()
                losses.append(loss.item())
        route = net.__dict__.get("_glam_graphed_route")
        if routed:
            assert route is not None and route.graphs() == 6, "three contents: a forward and a backward graph each"
        else:
            assert route is None or route.graphs() == 0
        results.append((losses, [p.detach().clone() for p in net.parameters()]))
    (l_e, p_e), (l_g, p_g) = results
    assert l_e == l_g, (l_e, l_g)
    for a, r in zip(p_g, p_e):
        assert torch.equal(a, r)


def test_same_shapes_different_content_are_different_graphs(device):
    """Two batches with identical shapes but different bonds must not share a graph (the fingerprint, not the shape, is the key), and a
    model in eval mode under no_grad replays a forward-only graph."""
    torch.manual_seed(3)
    net = _net(device).eval()
    a = synth_batch(12, seed=5).to(device)
    b = _fresh(a)
    perm = torch.randperm(b.edge_index.size(1), device=device)
    b.edge_index = b.edge_index.flip(0)[:, perm].contiguous()        # reversed and permuted bonds: same shapes, other graph structure
    b.edge_attr = b.edge_attr[perm].roll(1, 1).contiguous()
    net.graphed_call = False
    with torch.no_grad():
        ref_a, ref_b = net(a).clone(), net(b).clone()
    assert not torch.equal(ref_a, ref_b)
    net.graphed_call = True
    with torch.no_grad():
        for _ in range(4):
            assert torch.equal(net(_fresh(a)), ref_a) and torch.equal(net(_fresh(b)), ref_b)
    route = net.__dict__["_glam_graphed_route"]
    assert route.graphs() == 2
    # features are data, not identity: new x through the captured graph
    c = _fresh(a)
    c.x = torch.randn_like(c.x)
    net.graphed_call = False
    with torch.no_grad():
        ref_c = net(c).clone()
    net.graphed_call = True
    with torch.no_grad():
        assert torch.equal(net(_fresh(c)), ref_c) and route.graphs() == 2


def test_accumulating_gradients_and_switches(device, monkeypatch):
    """``zero_grad(set_to_none=False)`` (p.grad is the buffer autograd took over from the route: it must not be added to itself), two
    backward passes accumulating into p.grad, the per-model and the process-wide switch, and a knob flipped between two calls (a graph
    bakes the route in: the key carries the switches)."""
    from glam_amd import ops
    torch.manual_seed(4)
    net = _net(device)
    b = synth_batch(10, seed=8).to(device)
    b2 = _fresh(b)
    b2.x = torch.randn_like(b.x)                                        # the same graphs, other features: another gradient through the same key
    ref = copy.deepcopy(net)
    ref.graphed_call = False
    for step in range(5):
        for m in (net, ref):
            if step == 0:
                m.zero_grad(set_to_none=True)
            else:
                for p in m.parameters():
                    p.grad.zero_()
            _loss(m(_fresh(b)), b).backward()
            _loss(m(_fresh(b2)), b).backward()                          # accumulates into the p.grad the first backward left
        for p, q in zip(net.parameters(), ref.parameters()):
            assert torch.equal(p.grad, q.grad), step
    route = net.__dict__["_glam_graphed_route"]
    n = route.graphs()
    assert n == 2
    monkeypatch.setattr(ops, "GRU_WGRAD_BATCH", not ops.GRU_WGRAD_BATCH)
    net.zero_grad(set_to_none=True)
    _loss(net(_fresh(b)), b).backward()                                 # other switches: a new content key, eager again
    assert route.graphs() == n
    monkeypatch.undo()
    monkeypatch.setattr(graphs, "GRAPHED_CALL", False)
    before = route._states[next(iter(route._states))].visits
    net(_fresh(b))
    assert route._states[next(iter(route._states))].visits == before    # bypassed entirely
    monkeypatch.undo()
    # deepcopy / state_dict never see the route
    assert "_glam_graphed_route" not in net.state_dict() and copy.deepcopy(net).__dict__["_glam_graphed_route"].graphs() == 0


def test_training_mode_randomness_advances_between_replays(device):
    """RReLU / Dropout draw from a device-side Philox stream: two replays of the same training-mode graph differ, and the backward
    replay leaves finite gradients on every parameter."""
    torch.manual_seed(9)
    net = model.Architecture(mol_block="_TripletMessage", message_steps=2, e_dim=64).to(device).train()        # RReLU x 3, Dropout(0.2)
    b = synth_batch(10, seed=2).to(device)
    outs = [net(_fresh(b)).detach().clone() for _ in range(5)]
    route = net.__dict__["_glam_graphed_route"]
    assert route.graphs() == 2
    assert not torch.equal(outs[3], outs[4]), "replays of a training-mode graph must draw new masks"
    net.zero_grad(set_to_none=True)
    out = net(_fresh(b))
    out.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


def test_two_input_models_take_the_route_too(device):
    """``ArchitectureDTI`` (ligand + protein) in the same unchanged loop: route on vs off over five visits of two pair batches — the same
    trajectory within fp32 rounding (above 512 nodes a captured step sums the weight gradients of all applications of a block in one
    product, the eager one per application), with graphs replayed."""
    from glam_amd.data import synth_protein_batch
    from tests.conftest import assert_close
    torch.manual_seed(11)
    net0 = model.ArchitectureDTI(graph_do="_None()", end_do="_None()", pre_act="ReLU", graph_act="ReLU", flat_act="ReLU", end_act="ReLU", e_dim=64).to(device)
    pairs = [(synth_batch(4, seed=s).to(device), synth_protein_batch(4, seed=s + 10, n_min=40, n_max=90).to(device)) for s in (1, 2)]
    ys = [torch.randn(4, device=device) for _ in pairs]
    results = []
    for routed in (False, True):
        net = copy.deepcopy(net0)
        net.graphed_call = routed
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        losses = []
        for _epoch in range(5):
            for (mol, pro), y in zip(pairs, ys):
                opt.zero_grad()
                loss = torch.nn.functional.mse_loss(net(_fresh(mol), _fresh(pro)).view(-1), y)
                loss.backward()
                opt.step()
                losses.append(loss.item())
        if routed:
            assert net.__dict__["_glam_graphed_route"].graphs() == 4
        results.append((losses, [p.detach().clone() for p in net.parameters()]))
    (l_e, p_e), (l_g, p_g) = results
    assert np.allclose(l_e, l_g, rtol=2e-5, atol=1e-6), (l_e, l_g)
    for a, r in zip(p_g, p_e):
        assert_close(a, r, 2e-5, "parameter")


def _train_steps(net, opt, b, n):
    losses = []
    for _ in range(n):
        opt.zero_grad()
        loss = _loss(net(_fresh(b)), b)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    return losses


@pytest.mark.parametrize("trip", ["cpu_cuda", "double_float"])
def test_moved_parameter_storage_drops_the_captures(device, trip):
    """``model.cpu(); model.cuda()`` / ``model.double(); model.float()`` keep the Parameter OBJECTS and swap their storage (the
    ``best = copy.deepcopy(model.cpu()); model.cuda()`` idiom of a checkpointing trainer): graphs captured before the round trip read the
    old addresses and must be dropped — the routed trajectory stays the eager one to the bit, and the weights the graphs read are the
    ones the optimizer writes."""
    torch.manual_seed(12)
    net0 = _net(device)
    b = synth_batch(9, seed=31).to(device)
    results = []
    for routed in (False, True):
        net = copy.deepcopy(net0)
        net.graphed_call = routed
        opt = torch.optim.SGD(net.parameters(), lr=1e-4)     # (no per-parameter state tied to the old storage)
        losses = _train_steps(net, opt, b, 5)
        if routed:
            assert net.__dict__["_glam_graphed_route"].graphs() == 2
        ptr = next(net.parameters()).data_ptr()
        ident = [id(p) for p in net.parameters()]
        if trip == "cpu_cuda":
            net.cpu()
            net.to(device)
        else:
            net.double()
            net.float()
        assert [id(p) for p in net.parameters()] == ident, "the premise: the same Parameter objects"
        if routed:          # (the captures' proxies pin the old storage, so the new one is elsewhere; unrouted, the allocator may hand the block back)
            assert next(net.parameters()).data_ptr() != ptr, "the premise: new storage"
        losses += _train_steps(net, opt, b, 5)
        if routed:
            assert net.__dict__["_glam_graphed_route"].graphs() == 2, "captured again on the new storage"
        with torch.no_grad():
            losses.append(_loss(net(_fresh(b)), b).item())
        results.append((losses, [p.detach().clone() for p in net.parameters()]))
    (l_e, p_e), (l_g, p_g) = results
    assert l_e == l_g, (l_e, l_g)
    assert l_e[5] != l_e[4], "training went on after the round trip"
    for a, r in zip(p_g, p_e):
        assert torch.equal(a, r)


def test_freeze_then_unfreeze_after_capture(device):
    """A fine-tuning schedule: the message-passing blocks frozen for the first epochs, unfrozen later — on the same Parameter objects,
    after graphs were captured with the smaller trainable set.  The newly trainable parameters must receive gradients (eager's, to the
    bit) on already-captured batches."""
    torch.manual_seed(13)
    net0 = _net(device)
    b = synth_batch(9, seed=33).to(device)
    results = []
    for routed in (False, True):
        net = copy.deepcopy(net0)
        net.graphed_call = routed
        backbone = [p for n, p in net.named_parameters() if n.startswith("mol_conv")]
        assert backbone
        for p in backbone:
            p.requires_grad_(False)
        grads = []
        for phase in range(3):                       # frozen, unfrozen, frozen again
            for p in backbone:
                p.requires_grad_(phase == 1)
            for _ in range(4):
                net.zero_grad(set_to_none=True)
                _loss(net(_fresh(b)), b).backward()
            grads.append([None if p.grad is None else p.grad.clone() for p in net.parameters()])
            if routed:
                assert net.__dict__["_glam_graphed_route"].graphs() == 2, phase
        results.append(grads)
    for ge, gg in zip(*results):
        for a, r in zip(gg, ge):
            assert (a is None) == (r is None)
            assert a is None or torch.equal(a, r)
    assert all(g is not None for g in results[1][1]) and any(g is None for g in results[1][0]) and any(g is None for g in results[1][2])


def test_two_forwards_of_one_batch_before_the_first_backward(device):
    """A consistency loss: the same graph structure with two feature tensors, both outputs back-propagated afterwards.  A captured graph
    has ONE set of static activations — the second forward must not overwrite what the first output's backward needs."""
    torch.manual_seed(14)
    net = _net(device)
    ref = copy.deepcopy(net)
    ref.graphed_call = False
    b = synth_batch(9, seed=35).to(device)
    b2 = _fresh(b)
    b2.x = torch.randn_like(b.x)
    for _ in range(4):                               # (captured from the third visit)
        net.zero_grad(set_to_none=True)
        _loss(net(_fresh(b)), b).backward()
    assert net.__dict__["_glam_graphed_route"].graphs() == 2
    for m in (net, ref):
        m.zero_grad(set_to_none=True)
        o1 = m(_fresh(b))
        o2 = m(_fresh(b2))                           # same content key, o1's backward still pending
        (_loss(o1, b) + 0.5 * _loss(o2, b) + (o1 - o2).pow(2).mean()).backward()
    # (one backward over two graphs: autograd adds the four per-application gradients of a shared block one by one in the eager run, the
    #  replayed graph hands over its two already summed — the same numbers in another association: rounding-level agreement, not bits)
    from tests.conftest import assert_close
    for (n, p), q in zip(net.named_parameters(), ref.parameters()):
        assert_close(p.grad, q.grad, 2e-6, f"grad {n}")
    # the wrong answer this guards against — the first output's backward run on the second forward's activations — is far outside that
    o1 = ref(_fresh(b))
    assert (o1 - ref(_fresh(b2))).abs().max() > 1e-2
    # ... and the route is back on replays once nothing is pending
    st = next(iter(net.__dict__["_glam_graphed_route"]._states.values()))
    gen = st.gen
    net.zero_grad(set_to_none=True)
    _loss(net(_fresh(b)), b).backward()
    assert st.gen == gen + 1
