"""GPU (MI355X): the HIP path, called through the C ABI, against (1) the golden vectors captured
from the reference, (2) the CPU oracle on fresh seeds, (3) size-independent properties at the
BASELINE.json batch size.  fp32 tolerance: 1e-5 (scaled by max(1,|ref|)) as BASELINE.json states;
gradients 2e-5..5e-5 where several layers stack."""
import numpy as np
import os

import pytest
import torch

import oracle.glam_oracle as O
from glam_amd import layer, model, ops
from glam_amd.data import Data, synth_batch, synth_protein_batch
from tests.conftest import Golden, golden_names, assert_close, assert_fp32_parity, assert_golden_parity, assert_twin_parity

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _grads(out, cot, tensors):
    gs = torch.autograd.grad((out * cot).sum(), tensors, allow_unused=True)
    return [torch.zeros_like(t) if g is None else g for g, t in zip(gs, tensors)]


def _dev(d, device):
    return {k: v.to(device) for k, v in d.items()}


# ---------------------------------------------------------------------------------------------
# CSR staging
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["esol", "edge", "protein", "empty"])
def test_csr_matches_stable_sort(device, case):
    if case == "esol":
        b = synth_batch(200, seed=3)
        ei, N = b.edge_index, b.x.size(0)
    elif case == "edge":
        g = Golden("triplet_edge")
        ei, N = g.inputs["edge_index"], g.inputs["x"].size(0)
    elif case == "protein":
        b = synth_protein_batch(3, seed=5)
        ei, N = b.edge_index, b.x.size(0)
    else:
        ei, N = torch.zeros(2, 0, dtype=torch.long), 7
    gi = ops.GraphIndex(ei.to(device), N)
    colptr, dstv, eid_t = gi.transpose()
    for key_row, val_row, (rp, nb, ed) in [(1, 0, (gi.rowptr, gi.src, gi.eid)), (0, 1, (colptr, dstv, eid_t))]:
        key, val = ei[key_row].numpy(), ei[val_row].numpy()
        order = np.argsort(key, kind="stable")
        ref_ptr = np.concatenate([[0], np.cumsum(np.bincount(key, minlength=N))])
        assert np.array_equal(rp.cpu().numpy(), ref_ptr)
        assert np.array_equal(ed.cpu().numpy(), order)
        assert np.array_equal(nb.cpu().numpy(), val[order])


def test_csr_rejects_out_of_range_ids(device):
    ei = torch.tensor([[0, 1, 5], [1, 0, 2]], device=device)
    with pytest.raises(IndexError):
        ops.GraphIndex(ei, 3)
    with pytest.raises(IndexError):
        ops.SegmentPtr(torch.tensor([0, 2, 1], device=device), 3)


def test_deferred_id_validation_reports_late_but_never_loses_the_error(device, monkeypatch):
    """GLAM_VALIDATE=deferred: staging a foreign tensor does not sync; a raised flag surfaces at check_pending() (or at the next
    staging call once the copy has landed); valid tensors stay silent; 'off' never looks; bad edges are dropped on the device."""
    monkeypatch.setattr(ops, "VALIDATE", "deferred")
    ops.check_pending()
    good = torch.tensor([[0, 1, 2], [1, 0, 1]], device=device)
    ops.GraphIndex(good, 3)
    ops.check_pending()                                         # nothing to report
    bad = torch.tensor([[0, 1, 5], [1, 0, 2]], device=device)
    gi = ops.GraphIndex(bad, 3)                                 # no exception here
    assert gi.rowptr.tolist() == [0, 1, 2, 2], "the out-of-range edge is dropped, the rest is staged"
    with pytest.raises(IndexError, match="deferred"):
        ops.check_pending()
    ops.check_pending()                                         # reported once
    ops.SegmentPtr(torch.tensor([0, 2, 1], device=device), 3)
    torch.cuda.synchronize()
    with pytest.raises(IndexError, match="batch must be non-decreasing"):
        ops.GraphIndex(good.clone(), 3)                         # the next staging call polls
    # a whole model pass over a foreign (unmarked) batch: no flag, no exception, same numbers as the validated path
    b = synth_batch(16, seed=3).to(device)
    plain = type(b)()
    for k in ("x", "edge_index", "edge_attr", "batch", "y"):
        setattr(plain, k, getattr(b, k).clone())
    plain.num_graphs = b.num_graphs
    torch.manual_seed(0)
    net = model.Architecture(mol_in_dim=15, mol_edge_in_dim=4, mol_block="_TripletMessage", message_steps=2).to(device).eval()
    out_deferred = net(plain)
    ops.check_pending()
    monkeypatch.setattr(ops, "VALIDATE", "sync")
    assert torch.equal(net(b), out_deferred)
    monkeypatch.setattr(ops, "VALIDATE", "off")
    ops.GraphIndex(bad.clone(), 3)
    ops.check_pending()


def test_segment_ptr(device):
    batch = torch.tensor([0, 0, 2, 2, 2, 5], device=device)
    sp = ops.SegmentPtr(batch)
    assert sp.B == 6 and sp.ptr.tolist() == [0, 2, 2, 5, 5, 5, 6]


# ---------------------------------------------------------------------------------------------
# TripletMessage / TripletMessageLight against the golden vectors
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", golden_names("triplet_"))
def test_triplet_message_golden(device, name):
    g = Golden(name)
    conv = layer.TripletMessage(g.meta["C"], g.meta["De"]).to(device)
    conv.load_state_dict(g.params)
    x = g.inputs["x"].to(device).requires_grad_(True)
    ea = g.inputs["edge_attr"].to(device).requires_grad_(True)
    out = conv(x, g.inputs["edge_index"].to(device), ea)
    assert_close(out, g.out, TOL, name)
    names = [n for n, _ in conv.named_parameters()]
    gs = _grads(out, g.cot.to(device), [x, ea] + [p for _, p in conv.named_parameters()])

    def run64():
        xo, eo = g.inputs["x"].double().requires_grad_(True), g.inputs["edge_attr"].double().requires_grad_(True)
        ps = [g.params[n].double().requires_grad_(True) for n in names]
        o = O.triplet_message(xo, g.inputs["edge_index"], eo, *ps)
        return o, _grads(o, g.cot.double(), [xo, eo] + ps)
    assert_golden_parity(run64, g, out, gs, name, ["x", "edge_attr"] + names)


@pytest.mark.parametrize("ext", [False, True])
@pytest.mark.parametrize("name", golden_names("triplet_"))
def test_triplet_message_golden_inference_forward(device, name, ext, monkeypatch):
    """``torch.no_grad()`` (the evaluation passes of src_1gp/trainer.py:306-327): the forward that keeps nothing for a backward — no
    ``aggr``, no ``stats`` store — gives the training forward's output bit for bit and the golden one within the bound, through the Python
    node and through the torch-extension operator; ``INFER_FWD = False`` is the A/B route."""
    g = Golden(name)
    conv = layer.TripletMessage(g.meta["C"], g.meta["De"]).to(device)
    conv.load_state_dict(g.params)
    monkeypatch.setattr(ops, "USE_TORCH_EXT", ext)
    x, ei, ea = g.inputs["x"].to(device), g.inputs["edge_index"].to(device), g.inputs["edge_attr"].to(device)
    out_train = conv(x.clone().requires_grad_(True), ei, ea).detach()
    with torch.no_grad():
        out_inf = conv(x, ei, ea)
        monkeypatch.setattr(ops, "INFER_FWD", False)
        out_kept = conv(x, ei, ea)
    assert not out_inf.requires_grad
    assert torch.equal(out_inf, out_train) and torch.equal(out_kept, out_train)
    assert_close(out_inf, g.out, TOL, name + " (inference forward)")


def test_inference_forward_launches_and_c_abi(device, monkeypatch):
    """At the headline size: under ``torch.no_grad()`` the step launches the inference instantiation of the warp-specialised forward (its own
    label in the kernel trace), the general fused-update kernel takes NULL too where ``glam_triplet_layer_infer_supported`` says so, and
    through the C ABI ``aggr = stats = NULL`` leaves ``out`` bit-equal while a lone NULL is an argument error."""
    from glam_amd import _lib
    lib, p = _lib.load(), _lib.ptr
    b = synth_batch(1024, seed=0).to(device)
    torch.manual_seed(0)
    conv = layer.TripletMessage(60, 4).to(device)
    N = b.x.size(0)
    x = torch.randn(N, 60, device=device)
    monkeypatch.setattr(ops, "USE_TORCH_EXT", False)
    monkeypatch.setattr(ops, "INFER_FWD", True)
    outs, labels = {}, {}
    for route in ("auto", "0"):
        monkeypatch.setattr(ops, "WS_ROUTE", route)
        for grad in (True, False):
            with _lib.kernel_timer(capacity=16) as kt, (torch.enable_grad() if grad else torch.no_grad()):
                outs[route, grad] = conv(x.clone().requires_grad_(grad), b.edge_index, b.edge_attr).detach()
            labels[route, grad] = [n for n, _, _ in kt.records()]
        assert torch.equal(outs[route, True], outs[route, False]), route
    ws = lib.glam_triplet_layer_ws_supported(3, 60, 4, 1) == 1        # (GLAM_WS=0: the general kernels everywhere; their NULL form is below)
    if ws:
        assert any("k_triplet_fwd_ws<inference>" in n for n in labels["auto", False]), labels["auto", False]
    assert not any("inference" in n for n in labels["auto", True])
    assert lib.glam_triplet_layer_infer_supported(3, 60, 4) == 1
    # C ABI
    gi = ops.graph_index(b.edge_index, N)
    ell = gi.ell()
    with torch.no_grad():
        staged = torch.empty(lib.glam_triplet_staged_floats(3, 60, 4), device=device)
        assert lib.glam_triplet_stage_params(p(conv.weight_node), p(conv.weight_edge), p(conv.weight_triplet_att), p(conv.weight_scale), p(conv.bias),
                                             60, 3, 4, 60, 4, p(staged), _lib.stream()) == 0
    f = dict(dtype=torch.float32, device=device)
    res = []
    for keep in (True, False):
        xw, a_ij, out = torch.empty(N, 180, **f), torch.empty(N, 8, **f), torch.full((N, 60), float("nan"), **f)
        aggr, stats = (torch.empty(N, 180, **f), torch.empty(N, 8, **f)) if keep else (None, None)
        rc = lib.glam_triplet_layer_fwd_ell(p(x), p(b.edge_attr), p(staged), p(ell[0]), p(ell[1]), 1, N, gi.E, 3, 60, 4, 0.2, p(xw), p(a_ij), p(aggr),
                                            p(stats), p(out), _lib.stream())
        assert (rc == 0) == ws, lib.glam_last_error()
        if not ws:
            assert rc == _lib.GLAM_E_UNSUPPORTED
            out = outs["auto", True]
        res.append(out)
        out2 = torch.full((N, 60), float("nan"), **f)
        rc = lib.glam_triplet_layer_fwd(p(x), p(b.edge_attr), p(staged), p(gi.rowptr), p(gi.src), p(gi.eid), N, gi.E, 3, 60, 4, 0.2, p(xw), p(a_ij),
                                        p(aggr), p(stats), p(out2), _lib.stream())
        assert rc == 0, lib.glam_last_error()
        res.append(out2)
    assert torch.equal(res[0], res[2]) and torch.equal(res[1], res[3]) and torch.equal(res[0], outs["auto", True])
    assert lib.glam_triplet_layer_fwd_ell(p(x), p(b.edge_attr), p(staged), p(ell[0]), p(ell[1]), 1, N, gi.E, 3, 60, 4, 0.2, p(xw), p(a_ij), None,
                                          p(torch.empty(N, 8, **f)), p(out), _lib.stream()) != 0


@pytest.mark.parametrize("name", golden_names("triplet_"))
def test_triplet_aggregate_op_golden(device, name):
    """Op level: the fused kernel alone against the reference's aggregate (before ``update``)."""
    g = Golden(name)
    C, De = g.meta["C"], g.meta["De"]
    conv = layer.TripletMessage(C, De).to(device)
    conv.load_state_dict(g.params)
    x, ei, ea = g.inputs["x"].to(device), g.inputs["edge_index"].to(device), g.inputs["edge_attr"].to(device)
    with torch.no_grad():
        Wn, Wa, We, M, Ws, Cp, Dp = conv._staged_weights()
        ea_p = torch.nn.functional.pad(ea, (0, Dp - De))
        aggr = ops.triplet_aggregate(x @ Wn, x @ Wa, ea_p, We, M, ops.graph_index(ei, x.size(0)), 3, Cp)
    assert_close(aggr.view(-1, 3, Cp)[:, :, :C], g.grads["__aggr"], TOL, name + "/aggr")
    if Cp != C:
        assert (aggr.view(-1, 3, Cp)[:, :, C:] == 0).all()


@pytest.mark.parametrize("name", golden_names("light_"))
def test_triplet_light_golden(device, name):
    g = Golden(name)
    conv = layer.TripletMessageLight(g.meta["C"], g.meta["De"]).to(device)
    conv.load_state_dict(g.params)
    x = g.inputs["x"].to(device).requires_grad_(True)
    ea = g.inputs["edge_attr"].to(device).requires_grad_(True)
    out = conv(x, g.inputs["edge_index"].to(device), ea)
    assert_close(out, g.out, TOL, name)
    names = [n for n, _ in conv.named_parameters()]
    gs = _grads(out, g.cot.to(device), [x, ea] + [p for _, p in conv.named_parameters()])

    def run64():
        xo, eo = g.inputs["x"].double().requires_grad_(True), g.inputs["edge_attr"].double().requires_grad_(True)
        ps = [g.params[n].double().requires_grad_(True) for n in names]
        o = O.triplet_message_light(xo, g.inputs["edge_index"], eo, *ps)
        return o, _grads(o, g.cot.double(), [xo, eo] + ps)
    assert_golden_parity(run64, g, out, gs, name, ["x", "edge_attr"] + names)


# ---------------------------------------------------------------------------------------------
# readouts and norms
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", golden_names("pool5_"))
def test_pool5_golden(device, name):
    g = Golden(name)
    x = g.inputs["x"].to(device).requires_grad_(True)
    batch = g.inputs["batch"].to(device)
    out = layer.GlobalPool5()(x, batch)
    assert_close(out, g.out, TOL, name)
    assert_close(_grads(out, g.cot.to(device), [x])[0], g.grads["x"], TOL, name + "/gx")
    D = x.size(1)
    assert_close(layer.global_mean_pool(x, batch), g.out[:, :D], TOL, name + "/mean")
    assert_close(layer.global_add_pool(x, batch), g.out[:, D:2 * D], TOL, name + "/add")
    assert_close(layer.global_sort_pool(x, batch, 3), g.out[:, 2 * D:], TOL, name + "/sort")
    mx = layer.global_max_pool(x, batch)
    assert_close(mx, g.grads["__max"], 0, name + "/max")
    xo = g.inputs["x"].clone().requires_grad_(True)
    cot = torch.randn(mx.shape, generator=torch.Generator().manual_seed(1))
    (gref,) = _grads(O.global_max_pool(xo, g.inputs["batch"], g.meta["B"]), cot, [xo])
    assert_close(_grads(mx, cot.to(device), [x])[0], gref, 0, name + "/max.gx")


@pytest.mark.parametrize("name", golden_names("lapool_"))
def test_lapool_golden(device, name):
    g = Golden(name)
    pool = layer.GlobalLAPool(60).to(device)
    pool.load_state_dict(g.params)
    x = g.inputs["x"].to(device).requires_grad_(True)
    out = pool(x, g.inputs["batch"].to(device))
    assert_close(out, g.out, TOL, name)
    names = [n for n, _ in pool.named_parameters()]
    gs = _grads(out, g.cot.to(device), [x] + [p for _, p in pool.named_parameters()])

    def run64():
        xo = g.inputs["x"].double().requires_grad_(True)
        ps = [g.params[n].double().requires_grad_(True) for n in names]
        pd = dict(zip(names, ps))
        o = O.global_attention(xo, g.inputs["batch"], g.meta["B"], pd["pool.gate_nn.weight"], pd["pool.gate_nn.bias"], pd["pool.nn.weight"],
                               pd["pool.nn.bias"])
        return o, _grads(o, g.cot.double(), [xo] + ps)
    assert_golden_parity(run64, g, out, gs, name, ["x"] + names)


def test_set2set_golden(device):
    g = Golden("set2set_esol")
    s2s = layer.Set2Set(60, processing_steps=3).to(device)
    s2s.load_state_dict(g.params)
    x = g.inputs["x"].to(device).requires_grad_(True)
    out = s2s(x, g.inputs["batch"].to(device))
    assert_close(out, g.out, TOL, "set2set")
    names = [n for n, _ in s2s.named_parameters()]
    gs = _grads(out, g.cot.to(device), [x] + [p for _, p in s2s.named_parameters()])

    def run64():
        lstm = torch.nn.LSTM(120, 60).double()
        lstm.load_state_dict({k[5:]: v.double() for k, v in g.params.items()})
        xo = g.inputs["x"].double().requires_grad_(True)
        o = O.set2set(xo, g.inputs["batch"], g.meta["B"], lstm)
        return o, _grads(o, g.cot.double(), [xo] + [dict(lstm.named_parameters())[n[5:]] for n in names])
    assert_golden_parity(run64, g, out, gs, "set2set", ["x"] + names)


def test_norms_golden(device):
    g = Golden("norms_edge")
    x0, b = g.inputs["x"].to(device), g.inputs["batch"].to(device)
    cot = g.cot.to(device)
    for nm, fn in [("pair", lambda t: layer._PairNorm(60)(t, b)), ("pair_nobatch", lambda t: layer._PairNorm(60)(t, None)),
                   ("layer", lambda t: layer._LayerNorm(60).to(device)(t, b)),
                   ("gsize", lambda t: layer._GraphSizeNorm(60)(t, b))]:
        x = x0.clone().requires_grad_(True)
        out = fn(x)
        assert_close(out, g.grads[f"__out_{nm}"], TOL, nm)

        def run64(nm=nm):
            xo, bb, B = g.inputs["x"].double().requires_grad_(True), g.inputs["batch"], g.meta["B"]
            o = {"pair": lambda: O.pair_norm(xo, bb, B), "pair_nobatch": lambda: O.pair_norm(xo),
                 "layer": lambda: O.graph_layer_norm(xo, torch.ones(60, dtype=torch.float64), torch.zeros(60, dtype=torch.float64), bb, B),
                 "gsize": lambda: O.graph_size_norm(xo, None)}[nm]()
            return o, _grads(o, g.cot.double(), [xo])
        assert_golden_parity(run64, g, out, _grads(out, cot, [x]), nm, [f"__gx_{nm}"], out_key=f"__out_{nm}")


# ---------------------------------------------------------------------------------------------
# MessageBlock / Architecture / train step
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", golden_names("block_"))
def test_message_block_golden(device, name):
    g = Golden(name)
    m = g.meta
    blk = layer.MessageBlock(60, 60, 4, norm=m["norm"], dropout="_None()", conv=m["conv"], act=m["act"], res=True)
    blk.load_state_dict(g.params, strict=False)     # GATConv's shared lin_r alias is not in named_parameters()
    blk = blk.to(device).eval()
    i = _dev(g.inputs, device)
    x = i["x"].clone().requires_grad_(True)
    x1, h1 = blk(x, i["edge_index"], i["edge_attr"], h=None, batch=i["batch"])
    assert_close(x1, g.grads["__x1"], TOL, name + "/x1")
    x2, h2 = blk(x1, i["edge_index"], i["edge_attr"], h=h1, batch=i["batch"])
    assert_close(x2, g.out, TOL, name)
    assert_close(h2.squeeze(0), g.grads["__h"], TOL, name + "/h")
    names = [n for n, _ in blk.named_parameters()]
    gs = _grads(x2, g.cot.to(device), [x] + [p for _, p in blk.named_parameters()])

    def run64():
        xo = g.inputs["x"].double().requires_grad_(True)
        sd = {k: v.double().requires_grad_(True) for k, v in g.params.items()}
        ii = g.inputs
        o1, hh1 = O.message_block(sd, "", xo, ii["edge_index"], ii["edge_attr"].double(), None, ii["batch"], m["B"], m["conv"], m["norm"], m["act"])
        o2, _ = O.message_block(sd, "", o1, ii["edge_index"], ii["edge_attr"].double(), hh1, ii["batch"], m["B"], m["conv"], m["norm"], m["act"])
        return o2, _grads(o2, g.cot.double(), [xo] + [sd[n] for n in names])
    assert_golden_parity(run64, g, x2, gs, name, ["x"] + names)


@pytest.mark.parametrize("name", golden_names("arch_"))
def test_architecture_golden(device, name):
    g = Golden(name)
    m = g.meta
    net = model.Architecture(e_dim=m["e_dim"], out_dim=m["out_dim"], message_steps=m["message_steps"],
                             mol_block=m["mol_block"], mol_readout=m["mol_readout"], graph_norm=m.get("graph_norm", "_None"))
    net.load_state_dict(g.params)
    net = net.to(device).eval()
    i = _dev(g.inputs, device)
    data = Data(i["x"], i["edge_index"], i["edge_attr"], batch=i["batch"])
    data.num_graphs = m["B"]
    out = net(data)
    names = [n for n, _ in net.named_parameters()]
    gs = _grads(out, g.cot.to(device), [p for _, p in net.named_parameters()])
    # the golden is the reference's fp32 sample; the oracle's fp64 run on the same inputs gives its rounding-noise floor
    sd64 = {k: v.double().clone().requires_grad_(True) for k, v in g.params.items()}
    d64 = Data(g.inputs["x"].double(), g.inputs["edge_index"], g.inputs["edge_attr"].double(), batch=g.inputs["batch"])
    o64 = O.architecture(sd64, d64, m["B"], m["message_steps"], m["mol_block"], m["mol_readout"], graph_norm=m.get("graph_norm", "_None"))
    g64 = torch.autograd.grad((o64 * g.cot.double()).sum(), [sd64[n] for n in names], allow_unused=True)
    assert_fp32_parity(out, o64.detach(), g.out, name + " (golden)", out_tol=1e-5)
    for n, t, r64 in zip(names, gs, g64):
        if r64 is not None:
            # TripletMessageLight's attention vector gets its gradient through the separable form, d_att = W_node^T (x^T d_a): fp32
            # rounding acts on |W_node|^T |x^T d_a| (tens), not on |d_att| (0.1) — 3-4e-6 absolute whatever the upstream rounding sample
            # (measured with the fp32-MFMA and the 3 x bf16 input-gradient products alike); the reference sums d_logit * x_i directly
            k = 16 if ("Light" in m["mol_block"] and n.endswith("weight_triplet_att")) else 8
            assert_fp32_parity(t, r64, g.grads[n], f"{name}/grad.{n} (golden)", k=k)
    # three Adam steps exactly as TrainerMolRegression.train_iterations (trainer.py:286-298)
    net2 = model.Architecture(e_dim=m["e_dim"], out_dim=1, message_steps=m["message_steps"],
                              mol_block=m["mol_block"], mol_readout=m["mol_readout"], graph_norm=m.get("graph_norm", "_None"))
    net2.load_state_dict({k: (v if "lin_out1" not in k else v[:1]) for k, v in g.params.items()})
    net2 = net2.to(device).eval()
    opt = torch.optim.Adam(net2.parameters(), lr=1e-3)
    y = i["y"].view(-1)
    trace = []
    for _ in range(3):
        opt.zero_grad()
        loss = torch.nn.MSELoss()(net2(data).view(-1), y)
        loss.backward()
        gn = torch.sqrt(sum((p.grad ** 2).sum() for p in net2.parameters()))
        opt.step()
        trace.append([loss.item(), gn.item()])
    ref = g.grads["__train_trace"]
    assert_close(torch.tensor(trace), ref, 1e-4, name + "/train trace (loss, grad-norm)")
    assert_close(net2(data), g.grads["__post_out"], 1e-3, name + "/post-step output")


def test_dot_and_global_pool_golden(device):
    g = Golden("dotpool_pairs")
    i = _dev(g.inputs, device)
    mx, px = i["mol_x"].clone().requires_grad_(True), i["pro_x"].clone().requires_grad_(True)
    out2 = layer.dot_and_global_pool2(mx, px, i["mol_batch"], i["pro_batch"])
    assert_close(out2, g.out, TOL, "dot2")
    gm, gp = _grads(out2, g.cot.to(device), [mx, px])
    assert_close(gm, g.grads["mol_x"], TOL, "dot2 d_mol")
    assert_close(gp, g.grads["pro_x"], TOL, "dot2 d_pro")
    m5, p5 = i["mol_x"].clone().requires_grad_(True), i["pro_x"].clone().requires_grad_(True)
    out5 = layer.dot_and_global_pool5(m5, p5, i["mol_batch"], i["pro_batch"])      # [max, mean, median, min, std]
    assert_close(out5, g.grads["__out5"], TOL, "dot5")
    g5m, g5p = _grads(out5, g.grads["__cot5"].to(device), [m5, p5])

    def run64():
        mo, po = g.inputs["mol_x"].double().requires_grad_(True), g.inputs["pro_x"].double().requires_grad_(True)
        o = O.dot_and_global_pool(mo, po, g.inputs["mol_batch"], g.inputs["pro_batch"], 4, 5)
        return o, _grads(o, g.grads["__cot5"].double(), [mo, po])
    assert_golden_parity(run64, g, out5, [g5m, g5p], "dot5", ["__g5_mol", "__g5_pro"], out_key="__out5")


# ---------------------------------------------------------------------------------------------
# fresh seeds against the CPU oracle (sizes the oracle finishes in seconds)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C,De,kind", [(60, 4, "mol"), (30, 4, "mol"), (90, 4, "mol"), (60, 8, "protein"), (45, 8, "protein")])
def test_triplet_vs_oracle_fresh(device, C, De, kind):
    torch.manual_seed(100 + C + De)
    b = synth_batch(96, seed=C) if kind == "mol" else synth_protein_batch(3, seed=C, n_min=150, n_max=400)
    N = b.x.size(0)
    x0 = torch.randn(N, C)
    ea0 = b.edge_attr if kind == "mol" else torch.rand(b.edge_index.size(1), De)
    conv = layer.TripletMessage(C, De)
    with torch.no_grad():
        conv.bias.normal_(0, 0.1)
    ps0 = [p.detach().clone() for p in conv.parameters()]
    cot = torch.randn(N, C)

    def run(dt):
        xo = x0.to(dt).requires_grad_(True)
        ps = [p.to(dt).requires_grad_(True) for p in ps0]
        o = O.triplet_message(xo, b.edge_index, ea0.to(dt), *ps)
        return o, _grads(o, cot.to(dt), [xo] + ps)
    conv = conv.to(device)
    x = x0.to(device).requires_grad_(True)
    out = conv(x, b.edge_index.to(device), ea0.to(device))
    gs = _grads(out, cot.to(device), [x] + list(conv.parameters()))
    assert_twin_parity(run, out, gs, f"fresh {kind} C={C}", ["x"] + [n for n, _ in conv.named_parameters()])


@pytest.mark.parametrize("conv_name", ["_GCNConv", "_GATConv", "_NNConv", "_TripletMessageLight"])
def test_block_convs_run_and_are_graph_local(device, conv_name):
    """Every selectable conv (glam.py:63): outputs of graph g do not change when the other graphs
    of the batch are replaced (no edge crosses graphs)."""
    torch.manual_seed(5)
    b1, b2 = synth_batch(4, seed=1), synth_batch(4, seed=2)
    from glam_amd.data import Batch
    g0 = synth_batch(1, seed=9)
    h = torch.randn(g0.x.size(0), 30)

    def run(others):
        parts = [Data(h, g0.edge_index, g0.edge_attr)]
        n0 = 0
        for gidx in range(4):
            m = others.batch == gidx
            em = m[others.edge_index[0]]
            parts.append(Data(torch.randn(int(m.sum()), 30), others.edge_index[:, em] - n0, others.edge_attr[em]))
            n0 += int(m.sum())
        bb = Batch.from_data_list(parts).to(device)
        return blk(bb.x, bb.edge_index, bb.edge_attr, batch=bb.batch)[0][: h.size(0)]

    blk = layer.MessageBlock(30, 30, 4, norm="_None", dropout="_None()", conv=conv_name, act="ReLU", res=True).to(device).eval()
    a, c = run(b1), run(b2)
    assert torch.isfinite(a).all()
    assert_close(a, c, 1e-6, conv_name + " graph locality")


# ---------------------------------------------------------------------------------------------
# properties at the BASELINE.json size (B=1024 ESOL-shaped, C=60, H=3)
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def big(device):
    b = synth_batch(1024, seed=0)
    torch.manual_seed(0)
    conv = layer.TripletMessage(60, 4).to(device)
    x = torch.randn(b.x.size(0), 60, device=device)
    return b.to(device), conv, x


def test_full_size_determinism_and_permutation_invariance(big, device):
    b, conv, x = big
    x = x.clone().requires_grad_(True)
    out1 = conv(x, b.edge_index, b.edge_attr)
    (g1,) = torch.autograd.grad(out1.sum(), [x])
    x2 = x.detach().clone().requires_grad_(True)
    out2 = conv(x2, b.edge_index.clone(), b.edge_attr)
    (g2,) = torch.autograd.grad(out2.sum(), [x2])
    assert torch.equal(out1, out2) and torch.equal(g1, g2), "CSR segmented reduction must be bit-reproducible"
    perm = torch.randperm(b.edge_index.size(1), device=device)
    out3 = conv(x2, b.edge_index[:, perm].contiguous(), b.edge_attr[perm].contiguous())
    assert_close(out3, out1, TOL, "edge-order permutation invariance")


def test_full_size_softmax_partition_of_unity(big, device):
    """With W_edge rows equal (e_ij constant c) and xw constant 1, aggr = c * sum(alpha) = c for every
    node with an incoming edge, whatever the logits are."""
    b, conv, x = big
    N, E = x.size(0), b.edge_index.size(1)
    gi = ops.graph_index(b.edge_index, N)
    xw = torch.ones(N, 180, device=device)
    a_ij = torch.randn(N, 8, device=device) * 3
    M = torch.randn(4, 4, device=device)
    We = torch.full((4, 180), 0.5, device=device)
    aggr = ops.triplet_aggregate(xw, a_ij, b.edge_attr, We, M, gi, 3, 60)
    assert_close(aggr, torch.full_like(aggr, 0.5), 1e-6, "sum(alpha) == 1")


def test_full_size_linearity_in_xw(big, device):
    b, conv, x = big
    N = x.size(0)
    gi = ops.graph_index(b.edge_index, N)
    with torch.no_grad():
        Wn, Wa, We, M, Ws, Cp, Dp = conv._staged_weights()
        a_ij = x @ Wa
        u, v = torch.randn(N, 180, device=device), torch.randn(N, 180, device=device)
        f = lambda t: ops.triplet_aggregate(t, a_ij, b.edge_attr, We, M, gi, 3, 60)
        # (a property of the kernel, not a parity bound: two roundings per term over <= 4 neighbour terms of magnitude <= ~10)
        assert_close(f(2.0 * u + v), 2.0 * f(u) + f(v), 256 * 2.0 ** -23, "aggregate is linear in xw for fixed logits")


def test_full_size_vs_oracle_sample(big, device):
    """Whole B=1024 batch through the HIP layer; the oracle recomputes the first 64 graphs (graphs are
    independent) and must agree on those rows."""
    b, conv, x = big
    out = conv(x, b.edge_index, b.edge_attr)
    n64 = int((b.batch < 64).sum())
    em = (b.edge_index[0] < n64)
    ps = [p.detach().cpu() for p in conv.parameters()]
    ref = O.triplet_message(x[:n64].cpu(), b.edge_index[:, em].cpu(), b.edge_attr[em].cpu(), *ps)
    assert_close(out[:n64], ref, TOL, "first 64 graphs of the B=1024 batch")


def test_full_size_pool5_checksum(big, device):
    b, conv, x = big
    out = layer.GlobalPool5()(x, b.batch, 1024)
    assert_close(out[:, 60:120].sum(0), x.sum(0), 1e-5 * x.size(0) ** 0.5, "sum of per-graph sums == global sum")
    cnt = torch.bincount(b.batch, minlength=1024).view(-1, 1).float()
    assert_close(out[:, :60] * cnt, out[:, 60:120], 1e-5, "mean * count == sum")
    top = out[:, 120:].view(1024, 3, 60)[:, :, -1]
    assert (top[:, 0] >= top[:, 1]).all() and (top[:, 1] >= top[:, 2]).all(), "sort-pool rows are sorted"


# ---------------------------------------------------------------------------------------------
# dense ends: fp32-MFMA tall-skinny GEMM and weight-gradient GEMM against fp64 matmul
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,K1,K2,M1,M2,trans,bias", [
    (1000, 60, 0, 180, 8, 0, False), (1000, 180, 0, 60, 0, 0, True), (333, 60, 0, 180, 0, 1, False),
    (1, 180, 8, 60, 0, 1, False), (4099, 16, 0, 48, 8, 0, True), (17, 48, 8, 16, 0, 1, False), (0, 60, 0, 60, 0, 0, False),
    (5000, 64, 0, 184, 8, 0, True), (700, 188, 0, 64, 0, 0, False),
    # the 120 KB-image variant of the wide layers (K <= 96, M <= 320), and a launch with fewer items than wave slots
    (20400, 92, 0, 276, 8, 0, False), (900, 92, 0, 276, 0, 1, True), (3, 96, 0, 320, 0, 0, True), (40000, 180, 0, 60, 0, 0, True),
    # the long-reduction shapes beyond the fp32 table (K <= 288, M <= 96: tall_x3.hip, hid_dim_alpha = 6), both operand sources, ragged N
    (20400, 276, 8, 92, 0, 1, False), (900, 276, 0, 92, 0, 0, True), (5, 288, 0, 96, 0, 0, True), (3001, 200, 0, 64, 0, 0, False),
    (70000, 276, 8, 92, 0, 0, True), (20400, 180, 8, 60, 0, 1, True), (37, 100, 4, 8, 4, 0, True),
    # K <= 320, M <= 64 (NNConv's relation product 300 -> 60)
    (20400, 300, 0, 60, 0, 0, True), (1000, 320, 0, 64, 0, 1, False), (7, 292, 8, 12, 0, 0, True)])
def test_ts_gemm(device, N, K1, K2, M1, M2, trans, bias):
    from glam_amd import _lib
    lib, p = _lib.load(), _lib.ptr
    g = torch.Generator().manual_seed(N + K1 + M1)
    A1 = torch.randn(N, K1, generator=g)
    A2 = torch.randn(N, max(K2, 1), generator=g)[:, :K2].contiguous()
    K, M = K1 + K2, M1 + M2
    W = torch.randn(K, M, generator=g)
    b = torch.randn(M1, generator=g) if bias else None
    ref = torch.cat([A1, A2], 1).double() @ W.double()
    if bias:
        ref[:, :M1] += b.double()
    Wd = (W.t().contiguous() if trans else W).to(device)
    ldw = K if trans else M
    A1d, A2d = A1.to(device), A2.to(device)
    o1 = torch.full((N, M1), float("nan"), device=device)
    o2 = torch.full((N, max(M2, 1)), float("nan"), device=device)
    img = torch.empty(lib.glam_ts_gemm_image_bytes(K, M) // 4, device=device)
    assert lib.glam_ts_gemm_make_image(p(Wd), ldw, trans, K, M, p(img), _lib.stream()) == 0, lib.glam_last_error()
    bd = b.to(device) if bias else None
    rc = lib.glam_ts_gemm(p(A1d), K1, K1, p(A2d) if K2 else None, K2, K2, p(img), p(bd), p(o1), M1, M1,
                          p(o2) if M2 else None, M2, max(M2, 4), N, _lib.stream())
    assert rc == 0, lib.glam_last_error()
    assert_close(o1, ref[:, :M1], 2e-6, "ts_gemm out1")
    if M2:
        assert_close(o2, ref[:, M1:], 2e-6, "ts_gemm out2")


@pytest.mark.parametrize("N", [1, 15, 16, 17, 100, 640, 2047, 4113, 20400, 20401, 40800, 131071, 131072, 326411])
@pytest.mark.parametrize("bias,M2", [(True, 8), (False, 0)])
def test_node_gemm_w_once_per_block_equals_w_per_wave(device, monkeypatch, N, bias, M2):
    """The node GEMM x[N, 60] @ [W_node | Wa][60, 180 (+ 8)] (layer.py:37: ``self.weight_node``; k_ts_gemm_x3_sw: W split once per block
    into LDS, whole tiles and single column splits dealt over the waves) against k_ts_gemm_x3 (a W slice per wave, GLAM_TS_SW=0; its
    12-wave form from 131 072 rows): the same arithmetic in the same order, bit for bit — every unit mapping (whole tiles round the waves, 1..7 tiles left
    over, fewer tiles than blocks, a ragged last tile), with and without the second output and the bias; and against fp64."""
    from glam_amd import _lib
    lib, p = _lib.load(), _lib.ptr
    if not _lib.route_enabled("x3"):
        pytest.skip("GLAM_X3=0: both routes are the fp32 kernel")
    g = torch.Generator().manual_seed(N)
    K, M1 = 60, 180
    A = torch.randn(N, K, generator=g).to(device)
    W = torch.randn(K, M1 + M2, generator=g).to(device)
    b = torch.randn(M1, generator=g).to(device) if bias else None
    img = torch.empty(lib.glam_ts_gemm_image_bytes(K, M1 + M2) // 4, device=device)
    assert lib.glam_ts_gemm_make_image(p(W), M1 + M2, 0, K, M1 + M2, p(img), _lib.stream()) == 0, lib.glam_last_error()
    outs = []
    for sw in ("1", "0"):
        monkeypatch.setenv("GLAM_TS_SW", sw)
        o1 = torch.full((N, M1), float("nan"), device=device)
        o2 = torch.full((N, max(M2, 1)), float("nan"), device=device)
        rc = lib.glam_ts_gemm(p(A), K, K, None, 0, 0, p(img), p(b), p(o1), M1, M1, p(o2) if M2 else None, M2, max(M2, 4), N, _lib.stream())
        assert rc == 0, lib.glam_last_error()
        outs.append((o1, o2))
    assert torch.equal(outs[0][0], outs[1][0]) and (not M2 or torch.equal(outs[0][1], outs[1][1]))
    ref = A.double() @ W.double()
    if bias:
        ref[:, :M1] += b.double()
    assert_close(outs[0][0], ref[:, :M1], 2e-6, "node GEMM out1")
    if M2:
        assert_close(outs[0][1], ref[:, M1:], 2e-6, "node GEMM out2")


@pytest.fixture(params=["default", "x3"])
def wgrad_route(request, monkeypatch):
    """The weight-gradient products on both kernels: "default" = k_wgrad below 32 768 rows and k_wgrad_x3 (csrc/wgrad_x3.hip) from there,
    "x3" = k_wgrad_x3 at every size (GLAM_WGRAD_X3_ROWS is read at each launch)."""
    if request.param == "x3":
        monkeypatch.setenv("GLAM_WGRAD_X3_ROWS", "1")
    return request.param


@pytest.mark.parametrize("N,I1,I2,ones,J", [(1000, 180, 0, 1, 60), (20400, 180, 8, 0, 60), (7, 48, 8, 0, 16), (1, 16, 0, 1, 16), (5000, 56, 8, 1, 64), (3000, 184, 4, 1, 64),
                                            # 64 < J <= 128: two column chunks in one launch (wide layers)
                                            (20400, 276, 8, 0, 92), (5000, 276, 0, 1, 92), (33, 300, 16, 1, 128),
                                            # N >= 131072: the row-range form (k_wgrad_rows), 1..5 slabs, ragged last block
                                            (131072, 180, 8, 0, 64), (140001, 180, 0, 1, 60), (131075, 48, 8, 0, 16), (131100, 120, 0, 1, 32),
                                            (131073, 300, 16, 1, 64), (150000, 276, 8, 0, 92)])
def test_wgrad_gemm(device, N, I1, I2, ones, J, wgrad_route):
    from glam_amd import _lib
    lib, p = _lib.load(), _lib.ptr
    g = torch.Generator().manual_seed(N + I1)
    P1, P2, Q = torch.randn(N, I1, generator=g), torch.randn(N, max(I2, 1), generator=g)[:, :I2].contiguous(), torch.randn(N, J, generator=g)
    P = torch.cat([P1, P2] + ([torch.ones(N, 1)] if ones else []), 1)
    ref = P.double().t() @ Q.double()
    I = P.size(1)
    ws = torch.empty(lib.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=device)
    P1d, P2d, Qd = P1.to(device), P2.to(device), Q.to(device)
    for si, sj, shape in [(J, 1, (I, J)), (1, I, (J, I))]:
        out = torch.full(shape, float("nan"), device=device)
        rc = lib.glam_wgrad_gemm(p(P1d), I1, I1, p(P2d) if I2 else None, I2, I2, ones, p(Qd), J, J, 0, N, p(out), si, sj, p(ws),
                                 ws.numel(), _lib.stream())
        assert rc == 0, lib.glam_last_error()
        got = out if si == J else out.t()
        assert_close(got, ref, 3e-6 * max(1.0, N ** 0.5 / 10), "wgrad")


@pytest.mark.parametrize("N,I,ones,J", [(5000, 60, 1, 60), (3000, 300, 1, 60), (2000, 276, 0, 92), (777, 120, 1, 128), (131080, 60, 1, 64)])
def test_wgrad_gemm_add_sums_the_addend_in_the_reduction(device, N, I, ones, J, wgrad_route):
    """``glam_wgrad_gemm_add``: product + addend laid out like the output, in both stride orders, also through the two-chunk path
    (64 < J <= 128) and the large-N grid — exactly ``glam_wgrad_gemm``'s result plus the addend (one fp32 add per element)."""
    from glam_amd import _lib
    lib, p = _lib.load(), _lib.ptr
    g = torch.Generator().manual_seed(N + I + J)
    P, Q = torch.randn(N, I, generator=g).to(device), torch.randn(N, J, generator=g).to(device)
    It = I + ones
    ws = torch.empty(lib.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=device)
    for si, sj, shape in [(J, 1, (It, J)), (1, It, (J, It))]:
        add = torch.randn(shape, generator=g).to(device)
        base, out = torch.full(shape, float("nan"), device=device), torch.full(shape, float("nan"), device=device)
        assert lib.glam_wgrad_gemm(p(P), I, I, None, 0, 0, ones, p(Q), J, J, 0, N, p(base), si, sj, p(ws), ws.numel(), _lib.stream()) == 0
        rc = lib.glam_wgrad_gemm_add(p(P), I, I, None, 0, 0, ones, p(Q), J, J, 0, N, p(out), si, sj, p(add), p(ws), ws.numel(), _lib.stream())
        assert rc == 0, lib.glam_last_error()
        assert torch.equal(out, base + add)
    # a missing addend and an empty batch with one are argument errors, not launches
    assert lib.glam_wgrad_gemm_add(p(P), I, I, None, 0, 0, ones, p(Q), J, J, 0, N, p(out), si, sj, None, p(ws), ws.numel(), _lib.stream()) != 0
    assert lib.glam_wgrad_gemm_add(p(P), I, I, None, 0, 0, ones, p(Q), J, J, 0, 0, p(out), si, sj, p(add), p(ws), ws.numel(), _lib.stream()) != 0


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("graphs,nodes,D", [(12, 20, 60), (3, 300, 64), (5, 40, 45), (2, 700, 48)])
def test_graph_norm_bwd_add_c_abi(device, mode, graphs, nodes, D):
    """``glam_graph_norm_bwd_add`` in its three kernel forms (wave per graph, block per graph for hundreds of nodes, the generic
    odd-width one): exactly ``glam_graph_norm_bwd``'s result plus the addend."""
    from glam_amd import _lib
    lib, p = _lib.load(), _lib.ptr
    g = torch.Generator().manual_seed(graphs * 100 + D)
    sizes = torch.randint(max(1, nodes // 2), nodes + 1, (graphs,), generator=g)
    ptr_t = torch.zeros(graphs + 1, dtype=torch.int32)
    ptr_t[1:] = torch.cumsum(sizes, 0)
    N = int(ptr_t[-1])
    x, gy, add = (torch.randn(N, D, generator=g).to(device) for _ in range(3))
    ptr_d = ptr_t.to(device)
    base, out = torch.empty_like(x), torch.empty_like(x)
    assert lib.glam_graph_norm_bwd(p(x), p(gy), p(ptr_d), N, graphs, D, mode, 1.0, 1e-5, p(base), _lib.stream()) == 0
    rc = lib.glam_graph_norm_bwd_add(p(x), p(gy), p(ptr_d), N, graphs, D, mode, 1.0, 1e-5, p(add), p(out), _lib.stream())
    assert rc == 0, lib.glam_last_error()
    assert torch.equal(out, base + add)
    assert lib.glam_graph_norm_bwd_add(p(x), p(gy), p(ptr_d), N, graphs, D, mode, 1.0, 1e-5, None, p(out), _lib.stream()) != 0


# ---------------------------------------------------------------------------------------------
# edge cases of the boundary
# ---------------------------------------------------------------------------------------------
def test_no_edges_and_no_nodes(device):
    """A batch of single-atom molecules has E = 0: every output row is the bias (isolated nodes); N = 0 works too."""
    torch.manual_seed(3)
    conv = layer.TripletMessage(60, 4).to(device)
    with torch.no_grad():
        conv.bias.normal_()
    x = torch.randn(5, 60, device=device, requires_grad=True)
    ei = torch.zeros(2, 0, dtype=torch.long, device=device)
    ea = torch.zeros(0, 4, device=device)
    out = conv(x, ei, ea)
    assert_close(out, conv.bias.detach().expand(5, 60), 1e-7, "E=0 -> bias")
    (gx,) = torch.autograd.grad(out.sum(), [x])
    assert float(gx.abs().max()) == 0.0
    light = layer.TripletMessageLight(60, 4).to(device)
    assert_close(light(x, ei, ea), light.bias.detach().expand(5, 60), 1e-7, "light E=0 -> bias")
    out0 = conv(torch.zeros(0, 60, device=device), ei, ea)
    assert out0.shape == (0, 60)
    pooled = layer.GlobalPool5()(x, torch.tensor([0, 0, 2, 2, 2], device=device), 4)   # graphs 1 and 3 are empty
    assert pooled.shape == (4, 300) and float(pooled[1].abs().max()) == 0.0 and float(pooled[3].abs().max()) == 0.0


@pytest.mark.parametrize("heads,C,De", [(1, 60, 4), (2, 32, 1), (4, 44, 3)])
def test_other_head_counts_and_edge_widths(device, heads, C, De):
    torch.manual_seed(heads * 10 + De)
    b = synth_batch(40, seed=heads)
    N, E = b.x.size(0), b.edge_index.size(1)
    x0 = torch.randn(N, C)
    ea0 = torch.rand(E) if De == 1 else torch.rand(E, De)       # 1-D edge_attr takes the unsqueeze path (layer.py:39)
    conv = layer.TripletMessage(C, De, heads=heads)
    with torch.no_grad():
        conv.bias.normal_(0, 0.1)
    ps0 = [p.detach().clone() for p in conv.parameters()]
    cot = torch.randn(N, C)

    def run(dt):
        xo = x0.to(dt).requires_grad_(True)
        ps = [p.to(dt).requires_grad_(True) for p in ps0]
        o = O.triplet_message(xo, b.edge_index, ea0.view(E, De).to(dt), *ps, heads=heads)
        return o, _grads(o, cot.to(dt), [xo] + ps)
    conv = conv.to(device)
    x = x0.to(device).requires_grad_(True)
    out = conv(x, b.edge_index.to(device), ea0.to(device))
    assert_twin_parity(run, out, _grads(out, cot.to(device), [x] + list(conv.parameters())), f"heads={heads} C={C} De={De}",
                       ["x"] + [n for n, _ in conv.named_parameters()])


def test_hipgraph_capture_replays_identically(device):
    """Every entry point only enqueues on the current stream: a captured fwd+bwd step replays bit-identically."""
    torch.manual_seed(9)
    b = synth_batch(64, seed=4).to(device)
    conv = layer.TripletMessage(60, 4).to(device)
    x = torch.randn(b.x.size(0), 60, device=device, requires_grad=True)
    cot = torch.randn(b.x.size(0), 60, device=device)
    params = list(conv.parameters())

    def body():
        out = conv(x, b.edge_index, b.edge_attr)
        return [out] + list(torch.autograd.grad(out, params + [x], grad_outputs=cot))

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eager = [t.clone() for t in body()]          # also stages the CSR + transpose (one-time host sync)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        captured = body()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    for a, c in zip(eager, captured):
        assert torch.equal(a, c)


def test_edge_attr_gradient_optional_path(device):
    g = Golden("triplet_de8")
    conv = layer.TripletMessage(60, 8).to(device)
    conv.load_state_dict(g.params)
    ea = g.inputs["edge_attr"].to(device).requires_grad_(True)
    out = conv(g.inputs["x"].to(device), g.inputs["edge_index"].to(device), ea)
    (gea,) = torch.autograd.grad((out * g.cot.to(device)).sum(), [ea])

    def run64():
        names = ["weight_node", "weight_edge", "weight_triplet_att", "weight_scale", "bias"]
        eo = g.inputs["edge_attr"].double().requires_grad_(True)
        o = O.triplet_message(g.inputs["x"].double(), g.inputs["edge_index"], eo, *[g.params[n].double() for n in names])
        return o, _grads(o, g.cot.double(), [eo])
    assert_golden_parity(run64, g, None, [gea], "d_edge_attr via k_triplet_bwd_dea", ["edge_attr"])


@pytest.mark.parametrize("N,K,M,bias", [(1000, 60, 180, True), (777, 15, 60, True), (500, 180, 60, True), (64, 16, 8, False),
                                        (300, 57, 33, True), (1024, 300, 1024, True)])
def test_linear_op_matches_torch(device, N, K, M, bias):
    """ops.linear (k_ts_gemm / k_wgrad) against F.linear in fp64; the last shape is outside the kernel table and
    must route to the library GEMM."""
    g = torch.Generator().manual_seed(K * M)
    x0, w0 = torch.randn(N, K, generator=g), torch.randn(M, K, generator=g) / K ** 0.5
    b0 = torch.randn(M, generator=g) if bias else None
    cot = torch.randn(N, M, generator=g)
    xr, wr = x0.double().requires_grad_(True), w0.double().requires_grad_(True)
    br = b0.double().requires_grad_(True) if bias else None
    ref = torch.nn.functional.linear(xr, wr, br)
    gr = torch.autograd.grad((ref * cot.double()).sum(), [xr, wr] + ([br] if bias else []))
    x, w = x0.to(device).requires_grad_(True), w0.to(device).requires_grad_(True)
    b = b0.to(device).requires_grad_(True) if bias else None
    out = ops.linear(x, w, b)
    assert_close(out, ref, 3e-6, "linear fwd")
    gs = torch.autograd.grad((out * cot.to(device)).sum(), [x, w] + ([b] if bias else []))
    for n_, a, r in zip(["x", "w", "b"], gs, gr):
        assert_close(a, r, 1e-5, f"linear grad {n_}")


def test_gru_step_matches_torch_gru(device):
    torch.manual_seed(21)
    gru = torch.nn.GRU(60, 60)
    x0, h0 = torch.randn(500, 60), torch.randn(500, 60)
    xr, hr = x0.clone().requires_grad_(True), h0.clone().requires_grad_(True)
    out_ref, _ = gru(xr.unsqueeze(0), hr.unsqueeze(0))
    cot = torch.randn(500, 60)
    g_ref = torch.autograd.grad((out_ref.squeeze(0) * cot).sum(), [xr, hr] + list(gru.parameters()))
    gd = torch.nn.GRU(60, 60)
    gd.load_state_dict(gru.state_dict())
    gd = gd.to(device)
    x, h = x0.to(device).requires_grad_(True), h0.to(device).requires_grad_(True)
    out = ops.gru_step(x, h, gd.weight_ih_l0, gd.weight_hh_l0, gd.bias_ih_l0, gd.bias_hh_l0)
    assert_close(out, out_ref.squeeze(0), 2e-6, "gru fwd")
    gs = torch.autograd.grad((out * cot.to(device)).sum(), [x, h] + list(gd.parameters()))
    for n_, a, r in zip(["x", "h", "w_ih", "w_hh", "b_ih", "b_hh"], gs, g_ref):
        assert_close(a, r, 1e-5, f"gru grad {n_}")


@pytest.mark.parametrize("onehot", [True, False])
def test_nnconv_relation_and_general_paths(device, onehot):
    """NNConv(aggr='mean'): one-hot bonds take the relation-sum (R-GCN) kernel path, arbitrary edge features the
    per-edge-weight path; both against the oracle, forward and gradients."""
    torch.manual_seed(31)
    b = synth_batch(24, seed=8)
    N, E = b.x.size(0), b.edge_index.size(1)
    ea0 = b.edge_attr if onehot else torch.rand(E, 4)
    blk = layer._NNConv(32, 32, 4)
    x0 = torch.randn(N, 32)
    sd0 = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    names = list(sd0)
    cot = torch.randn(N, 32)

    def run(dt):
        sd = {k: v.to(dt).requires_grad_(True) for k, v in sd0.items()}
        xo = x0.to(dt).requires_grad_(True)
        o = O.nnconv_mean(xo, b.edge_index, ea0.to(dt), sd["conv.nn.0.weight"], sd["conv.nn.0.bias"], sd["conv.nn.2.weight"],
                          sd["conv.nn.2.bias"], sd["conv.root"], sd["conv.bias"])
        return o, _grads(o, cot.to(dt), [xo] + [sd[k] for k in names])
    blk = blk.to(device)
    x = x0.to(device).requires_grad_(True)
    out = blk(x, b.edge_index.to(device), ea0.to(device))
    params = dict(blk.named_parameters())
    assert_twin_parity(run, out, _grads(out, cot.to(device), [x] + [params[k] for k in names]), "nnconv", ["x"] + names)


def test_two_tower_model_runs(device):
    """ArchitectureDTI (src_2gi_dti_scr/model.py): ligand + protein towers with the per-pair fusion kernel."""
    torch.manual_seed(2)
    mb = synth_batch(6, seed=3).to(device)
    pb = synth_protein_batch(6, seed=4, n_min=60, n_max=120).to(device)
    net = model.ArchitectureDTI(mol_block="_TripletMessage", pro_block="_TripletMessage", e_dim=64, message_steps=2,
                                graph_do="_None()", end_do="_None()").to(device)
    out = net(mb, pb)
    assert out.shape == (6, 1) and torch.isfinite(out).all()
    out.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


def test_layer_parameter_gradients_form_one_bucket(device):
    from glam_amd.parallel import flat_view
    b = synth_batch(16, seed=1).to(device)
    conv = layer.TripletMessage(60, 4).to(device)
    x = torch.randn(b.x.size(0), 60, device=device)
    out = conv(x, b.edge_index, b.edge_attr)
    params = list(conv.parameters())
    grads = torch.autograd.grad(out.sum(), params)
    fv = flat_view(grads)
    assert fv is not None and fv.numel() == sum(p.numel() for p in params)
    assert torch.equal(fv, torch.cat([g.reshape(-1) for g in grads]))


@pytest.mark.parametrize("C,H,De", [(60, 3, 4), (30, 2, 4), (44, 4, 8)])
def test_layer_bwd_both_abi_routes_agree(device, C, H, De):
    """`glam_triplet_layer_bwd` + `glam_triplet_stage_params_bwd` (padded intermediate) and `glam_triplet_layer_bwd_params`
    (parameter gradients straight from the partial sums) are the same mathematics in two summation orders."""
    from glam_amd import _lib
    from glam_amd._lib import check, ptr, stream
    lib = _lib.load()
    torch.manual_seed(C + H)
    b = synth_batch(48, seed=C).to(device)
    N, E = b.x.size(0), b.edge_index.size(1)
    Cp, Dp = (C + 3) // 4 * 4, 4 if De <= 4 else 8
    HC = H * Cp
    f = dict(dtype=torch.float32, device=device)
    wn, we, att = torch.randn(C, H * C, **f) * 0.2, torch.randn(De, H * C, **f) * 0.2, torch.randn(H, 3 * C, **f) * 0.2
    wsc, bias = torch.randn(H * C, C, **f) * 0.2, torch.randn(C, **f) * 0.1
    x_p = torch.nn.functional.pad(torch.randn(N, C, **f), (0, Cp - C))
    ea_p = torch.nn.functional.pad(torch.rand(E, De, **f), (0, Dp - De))
    d_out = torch.nn.functional.pad(torch.randn(N, C, **f), (0, Cp - C))
    gi = ops.graph_index(b.edge_index, N)
    colptr, dst, eid_t = gi.transpose()
    staged = torch.empty(lib.glam_triplet_staged_floats(H, Cp, Dp), **f)
    check(lib.glam_triplet_stage_params(ptr(wn), ptr(we), ptr(att), ptr(wsc), ptr(bias), C, H, De, Cp, Dp, ptr(staged), stream()), "stage")
    xw, a_ij, aggr = torch.empty(N, HC, **f), torch.empty(N, 8, **f), torch.empty(N, HC, **f)
    stats, out = torch.empty(N, 8, **f), torch.empty(N, Cp, **f)
    check(lib.glam_triplet_layer_fwd(ptr(x_p), ptr(ea_p), ptr(staged), ptr(gi.rowptr), ptr(gi.src), ptr(gi.eid), N, E, H, Cp,
                                     Dp, 0.2, ptr(xw), ptr(a_ij), ptr(aggr), ptr(stats), ptr(out), stream()), "fwd")
    ws = torch.empty(lib.glam_triplet_layer_bwd_workspace_bytes(N, E, H, Cp, Dp), dtype=torch.uint8, device=device)

    def grads():
        return [torch.empty_like(wn), torch.empty_like(we), torch.empty_like(att), torch.empty_like(wsc), torch.empty_like(bias)]

    # route 1: padded intermediate + chain-rule kernel
    dstaged = torch.empty(lib.glam_triplet_dstaged_floats(H, Cp, Dp), **f)
    dx1, g1 = torch.empty_like(x_p), grads()
    check(lib.glam_triplet_layer_bwd(ptr(x_p), ptr(ea_p), ptr(staged), ptr(xw), ptr(a_ij), ptr(aggr), ptr(stats), ptr(d_out),
                                     ptr(gi.rowptr), ptr(gi.src), ptr(gi.eid), ptr(colptr), ptr(dst), ptr(eid_t), N, E, H, Cp, Dp, 0.2,
                                     ptr(dx1), ptr(dstaged), None, ptr(ws), ws.numel(), stream()), "bwd")
    check(lib.glam_triplet_stage_params_bwd(ptr(wn), ptr(we), ptr(att), ptr(dstaged), C, H, De, Cp, Dp, *[ptr(t) for t in g1], stream()),
          "stage_bwd")
    # route 2: one call
    dx2, g2 = torch.empty_like(x_p), grads()
    check(lib.glam_triplet_layer_bwd_params(ptr(x_p), ptr(ea_p), ptr(staged), ptr(xw), ptr(a_ij), ptr(aggr), ptr(stats), ptr(d_out),
                                            ptr(gi.rowptr), ptr(gi.src), ptr(gi.eid), ptr(colptr), ptr(dst), ptr(eid_t), N, E, C, H, De,
                                            Cp, Dp, 0.2, ptr(wn), ptr(we), ptr(att), ptr(dx2), *[ptr(t) for t in g2], None, ptr(ws),
                                            ws.numel(), stream()), "bwd_params")
    assert torch.equal(dx1, dx2)
    # ... and both sit inside the fp64-twin bound of the oracle on the same inputs (two fixed summation orders of one mathematics)
    twin = {}
    for dt in (torch.float32, torch.float64):
        ps = [t.detach().cpu().to(dt).requires_grad_(True) for t in (wn, we, att.view(1, H, 3 * C), wsc, bias)]
        o = O.triplet_message(x_p[:, :C].cpu().to(dt), b.edge_index.cpu(), ea_p[:, :De].cpu().to(dt), *ps, heads=H)
        twin[dt] = torch.autograd.grad((o * d_out[:, :C].cpu().to(dt)).sum(), ps)
    for name, a, c, r64, r32 in zip(["weight_node", "weight_edge", "att", "weight_scale", "bias"], g1, g2, twin[torch.float64], twin[torch.float32]):
        assert_fp32_parity(a.view(r64.shape), r64, r32, "route 1 " + name)
        assert_fp32_parity(c.view(r64.shape), r64, r32, "route 2 " + name)


@pytest.mark.parametrize("celu_in", [False, True])
@pytest.mark.parametrize("act,res", [("ReLU", True), ("LeakyReLU", True), ("CELU", False), ("_None", True)])
def test_gru_tail_matches_torch(device, act, res, celu_in):
    """GRU gates + residual + activation in one launch (layer.py:262-266) vs torch.nn.GRU + torch elementwise."""
    torch.manual_seed(3)
    N, C = 333, 60
    gru = torch.nn.GRU(C, C).to(device)
    xr = torch.randn(N, C, device=device, requires_grad=True)          # raw conv output
    x = torch.celu(xr) if celu_in else xr.clone().detach().requires_grad_(True)
    h = torch.randn(N, C, device=device, requires_grad=True)
    ident = torch.randn(N, C, device=device, requires_grad=True)
    cot_o, cot_h = torch.randn(N, C, device=device), torch.randn(N, C, device=device)
    fn = {"ReLU": torch.relu, "LeakyReLU": lambda t: torch.nn.functional.leaky_relu(t, 0.01), "CELU": torch.celu, "_None": lambda t: t}[act]
    y_ref, _ = gru(x.unsqueeze(0), h.unsqueeze(0))
    hn_ref = y_ref.squeeze(0)
    out_ref = fn(hn_ref + ident if res else hn_ref)
    tensors = [xr if celu_in else x, h, ident] + list(gru.parameters())
    g_ref = torch.autograd.grad([out_ref, hn_ref], tensors, grad_outputs=[cot_o, cot_h], allow_unused=True)
    code = {"ReLU": "relu", "LeakyReLU": "leaky", "CELU": "celu", "_None": "none"}[act]
    out, hn = ops.gru_tail(xr if celu_in else x, h, ident if res else None, gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0,
                           gru.bias_hh_l0, act=code, slope=0.01, celu_in=celu_in)
    g = torch.autograd.grad([out, hn], tensors, grad_outputs=[cot_o, cot_h], allow_unused=True)
    assert_close(out, out_ref, 2e-6, "out")
    assert_close(hn, hn_ref, 2e-6, "h_new")
    twin = {}
    for dt in (torch.float32, torch.float64):      # the same step through torch's CPU GRU in both precisions: the fp64 twin
        gc = torch.nn.GRU(C, C).to(dt)
        gc.load_state_dict({k: v.detach().cpu().to(dt) for k, v in gru.state_dict().items()})
        xr_c, h_c, id_c = (t.detach().cpu().to(dt).requires_grad_(True) for t in (xr, h, ident))
        x_c = torch.celu(xr_c) if celu_in else xr_c
        y_c, _ = gc(x_c.unsqueeze(0), h_c.unsqueeze(0))
        o_c = fn(y_c.squeeze(0) + id_c if res else y_c.squeeze(0))
        twin[dt] = torch.autograd.grad([o_c, y_c.squeeze(0)], [xr_c, h_c, id_c] + list(gc.parameters()),
                                       grad_outputs=[cot_o.cpu().to(dt), cot_h.cpu().to(dt)], allow_unused=True)
    for name, a, r, r64, r32 in zip(["x", "h", "identity", "w_ih", "w_hh", "b_ih", "b_hh"], g, g_ref, twin[torch.float64], twin[torch.float32]):
        if r is None:
            assert a is None, name
        else:
            assert_fp32_parity(a, r64, r32, f"grad.{name}")


def test_graphed_train_step_follows_the_eager_trajectory(device):
    """One hipGraph per cached batch (first visit eager, second captured, then replayed): same parameters as eager
    training after three epochs."""
    import copy
    from glam_amd.data import DataLoader, synth_molecule
    from glam_amd.graphs import GraphedTrainStep
    rng = np.random.default_rng(11)
    mols = [synth_molecule(rng) for _ in range(24)]
    torch.manual_seed(5)
    net0 = model.Architecture(mol_block="_TripletMessage", message_steps=2, mol_readout="GlobalPool5", e_dim=64, graph_norm="_None",
                              graph_do="_None()", end_do="_None()", pre_act="ReLU", graph_act="ReLU", flat_act="ReLU").to(device)
    loss_fn = lambda out, b: torch.nn.functional.mse_loss(out.view(-1), b.y.view(-1))
    results = []
    for graphed in (False, True):
        net = copy.deepcopy(net0)
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True)
        loader = DataLoader(mols, batch_size=8, device=device)          # cached batches: same objects every epoch
        stepper = GraphedTrainStep(net, opt, loss_fn)
        losses = []
        for _epoch in range(3):
            for b in loader:
                if graphed:
                    losses.append(float(stepper(b)))
                else:
                    opt.zero_grad(set_to_none=True)
                    loss = loss_fn(net(b), b)
                    loss.backward()
                    opt.step()
                    losses.append(float(loss))
        if graphed:
            assert stepper.graphs() == 3
        results.append((losses, [p.detach().clone() for p in net.parameters()]))
    (l_e, p_e), (l_g, p_g) = results
    assert np.allclose(l_e, l_g, rtol=1e-5, atol=1e-6), (l_e, l_g)
    for a, r in zip(p_g, p_e):
        assert_close(a, r, 1e-5, "parameter")


def test_graphed_train_step_follows_lr_schedulers(device):
    """The reference drives lr with ReduceLROnPlateau (src_1gp/trainer.py:55,85), which assigns a new FLOAT to
    param_group['lr'] between epochs: replayed graphs must step with the new rate (lr lives in a device tensor the captured
    optimizer launch reads), i.e. follow the eager trajectory through two reductions."""
    import copy
    from glam_amd.data import DataLoader, synth_molecule
    from glam_amd.graphs import GraphedTrainStep
    rng = np.random.default_rng(12)
    mols = [synth_molecule(rng) for _ in range(16)]
    torch.manual_seed(6)
    net0 = model.Architecture(mol_block="_TripletMessage", message_steps=2, mol_readout="GlobalPool5", e_dim=64, graph_norm="_None",
                              graph_do="_None()", end_do="_None()", pre_act="ReLU", graph_act="ReLU", flat_act="ReLU").to(device)
    loss_fn = lambda out, b: torch.nn.functional.mse_loss(out.view(-1), b.y.view(-1))
    results = []
    for graphed in (False, True):
        net = copy.deepcopy(net0)
        # learning rates that are powers of two: the float the eager optimizer multiplies by and the fp32 device tensor the
        # graphed one reads are then the same number, and the two trajectories can be compared bit for bit (Adam's
        # m / sqrt(v) amplifies a 1e-8 difference in lr to O(lr) on parameters whose gradient is ~0)
        opt = torch.optim.Adam(net.parameters(), lr=2.0 ** -7, capturable=True)
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.125, patience=0)
        loader = DataLoader(mols, batch_size=8, device=device)
        stepper = GraphedTrainStep(net, opt, loss_fn)
        lrs = []
        for epoch in range(5):
            for b in loader:
                if graphed:
                    stepper(b)
                else:
                    opt.zero_grad(set_to_none=True)
                    loss_fn(net(b), b).backward()
                    opt.step()
            sched.step(1.0)                      # a constant metric: "no improvement" from the second epoch on -> lr / 8 each time
            if epoch == 3:                       # and a plain float assignment, as older schedulers / user code do
                opt.param_groups[0]["lr"] = 2.0 ** -20
            lrs.append(float(opt.param_groups[0]["lr"]))
        results.append((lrs, [p.detach().clone() for p in net.parameters()]))
    (lr_e, p_e), (lr_g, p_g) = results
    assert lr_e == lr_g and lr_e[-1] < 1e-4 < lr_e[0], (lr_e, lr_g)
    for a, r in zip(p_g, p_e):
        assert_close(a, r, 1e-5, "parameter after lr reductions")
    with pytest.raises(ValueError):
        GraphedTrainStep(net0, torch.optim.Adam(net0.parameters(), lr=1e-3, fused=True), loss_fn)   # fused without capturable


def test_graphed_forward_tracks_parameter_updates(device):
    from glam_amd.graphs import GraphedForward
    torch.manual_seed(8)
    net = model.Architecture(mol_block="_TripletMessageLight", message_steps=2, mol_readout="GlobalLAPool", e_dim=32, graph_norm="_PairNorm",
                             pre_act="ReLU", graph_act="CELU", flat_act="ReLU").to(device).eval()
    b = synth_batch(12, seed=3).to(device)
    gf = GraphedForward(net)
    with torch.no_grad():
        ref0 = net(b)
    assert torch.equal(gf(b), ref0)                    # visit 1: eager
    assert torch.equal(gf(b).clone(), ref0)            # visit 2: captured + replayed
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(1.01)                               # "training" between evaluations
        ref1 = net(b)
    out1 = gf(b).clone()                               # visit 3: replay re-reads the updated parameters
    assert torch.equal(out1, ref1) and not torch.equal(out1, ref0)


@pytest.mark.parametrize("sizes", [[150, 90, 301, 64], [70, 500], [3, 260, 1, 2, 300]])
def test_pool5_large_graphs_block_kernel(device, sizes):
    """Protein-sized graphs take the block-per-graph GlobalPool5 kernels (N / B >= 64); ties on the sort channel included."""
    torch.manual_seed(sum(sizes))
    N, B, D = sum(sizes), len(sizes), 60
    batch = torch.repeat_interleave(torch.arange(B), torch.tensor(sizes))
    x0 = torch.randn(N, D)
    x0[5:9, D - 1] = 7.0                                  # ties among the top entries of graph 0: lower node index first
    x0[-1, D - 1] = x0[-2, D - 1]
    xo = x0.clone().requires_grad_(True)
    ref = O.global_pool5(xo, batch, B)
    cot = torch.randn(ref.shape)
    (g_ref,) = _grads(ref, cot, [xo])
    x = x0.to(device).requires_grad_(True)
    out = layer.GlobalPool5()(x, batch.to(device), B)

    def run(dt):
        xr = x0.to(dt).requires_grad_(True)
        o = O.global_pool5(xr, batch, B)
        return o.detach(), _grads(o, cot.to(dt), [xr])
    assert_twin_parity(run, out, _grads(out, cot.to(device), [x]), "pool5", ["x"])


@pytest.mark.parametrize("D", [15, 30, 45, 90, 92, 128])
def test_pool5_on_zero_padded_rows(device, D):
    """GlobalPool5 of the odd hidden widths (glam.py:60) reads the zero-padded rows the blocks hand on (ld = ceil4(D): 16 lanes per
    row up to 64, 32 up to 128) and writes the reference's compact [B, 5 D]; sort key = channel D - 1 (not the pad), ties and
    graphs shorter than k included.  Bit-equal to the same kernel family on the compact rows where both exist, oracle otherwise."""
    torch.manual_seed(D)
    sizes = [20, 1, 2, 33, 70, 3, 12]
    N, B, ld = sum(sizes), len(sizes), (D + 3) // 4 * 4
    batch = torch.repeat_interleave(torch.arange(B), torch.tensor(sizes))
    x0 = torch.randn(N, D)
    x0[22:25, D - 1] = 5.0                                 # ties among the top entries of graph 3
    cot = torch.randn(B, 5 * D)
    xp = torch.zeros(N, ld, device=device)
    xp[:, :D] = x0.to(device)
    xp.requires_grad_(True)
    v = ops.slice_cols(xp, D) if ld != D else xp
    out = layer.GlobalPool5()(v, batch.to(device), B)
    (g,) = _grads(out, cot.to(device), [xp])
    assert g.shape == (N, ld) and (ld == D or (g[:, D:] == 0).all()), "the gradient comes back in the padded layout, pads zero"

    def run(dt):
        xr = x0.to(dt).requires_grad_(True)
        o = O.global_pool5(xr, batch, B)
        return o.detach(), _grads(o, cot.to(dt), [xr])
    assert_twin_parity(run, out, [g[:, :D]], f"pool5 on padded rows, D = {D}", ["x"])
    # the C ABI directly: ld == D is the unpadded entry point; a padded call must agree with it exactly for ld <= 64
    raw = ops._lib.load()
    sp = ops.segment_ptr(batch.to(device), B)
    st = lambda: torch.cuda.current_stream().cuda_stream
    if ld != D and ld <= 64:
        xc = x0.to(device).contiguous()
        o1, t1 = torch.empty(B, 5 * D, device=device), torch.empty(B, 3, dtype=torch.int32, device=device)
        assert raw.glam_pool5_fwd(xc.data_ptr(), sp.ptr.data_ptr(), N, B, D, 3, o1.data_ptr(), t1.data_ptr(), st()) == 0
        o2, t2 = torch.empty_like(o1), torch.empty_like(t1)
        assert raw.glam_pool5_padded_fwd(xp.data_ptr(), sp.ptr.data_ptr(), N, B, ld, D, 3, o2.data_ptr(), t2.data_ptr(), st()) == 0
        assert torch.equal(t1, t2)
        assert_close(o2, o1, 1e-6, "padded vs compact rows")      # (different summation order: serial vs row groups)
    assert raw.glam_pool5_padded_fwd(xp.data_ptr(), sp.ptr.data_ptr(), N, B, ld + 4, D, 3, out.data_ptr(), sp.ptr.data_ptr(), st()) != 0
    assert raw.glam_pool5_padded_fwd(xp.data_ptr(), sp.ptr.data_ptr(), N, B, D - 1, D, 3, out.data_ptr(), sp.ptr.data_ptr(), st()) != 0


def test_pad_group_c_abi(device):
    """glam_pad_group: the zero-padded re-layouts of a module's parameters (GRU gate matrices [3C, C] -> [3Cp, Cp], biases,
    Linear weight [M, K] -> [Mp, Kp]) from one launch, their gradients back through one; against F.pad, bit for bit."""
    import ctypes
    import torch.nn.functional as F
    torch.manual_seed(5)
    C, Cp = 45, 48
    w1, w2, b1, b2 = (torch.randn(3 * C, C, device=device), torch.randn(3 * C, C, device=device), torch.randn(3 * C, device=device),
                      torch.randn(3 * C, device=device))
    wl, bl = torch.randn(30, 15, device=device), torch.randn(30, device=device)
    items = [(w1, (3, C, C), (Cp, Cp), (3 * Cp, Cp)), (w2, (3, C, C), (Cp, Cp), (3 * Cp, Cp)), (b1, (1, 3, C), (3, Cp), (3 * Cp,)),
             (b2, (1, 3, C), (3, Cp), (3 * Cp,)), (wl, (1, 30, 15), (32, 16), (32, 16)), (bl, (1, 1, 30), (1, 32), (32,))]
    for t, _, _, _ in items:
        t.requires_grad_(True)
    outs = ops.pad_group(items)
    want = [F.pad(w1.view(3, C, C), (0, 3, 0, 3)).reshape(3 * Cp, Cp), F.pad(w2.view(3, C, C), (0, 3, 0, 3)).reshape(3 * Cp, Cp),
            F.pad(b1.view(3, C), (0, 3)).reshape(-1), F.pad(b2.view(3, C), (0, 3)).reshape(-1), F.pad(wl, (0, 1, 0, 2)), F.pad(bl, (0, 2))]
    for o, w in zip(outs, want):
        assert o.shape == w.shape and torch.equal(o, w)
    cots = [torch.randn_like(o) for o in outs]
    srcs = [t for t, _, _, _ in items]
    # the gradient of the fourth output does not exist: zeros for that parameter, no error
    got = torch.autograd.grad([o for i, o in enumerate(outs) if i != 3], srcs, [c for i, c in enumerate(cots) if i != 3], allow_unused=True)
    ref = torch.autograd.grad([w for i, w in enumerate(want) if i != 3], srcs, [c for i, c in enumerate(cots) if i != 3], allow_unused=True)
    for i, (g, r) in enumerate(zip(got, ref)):
        assert torch.equal(g, r if r is not None else torch.zeros_like(srcs[i])), i
    raw = ops._lib.load()
    vp = ctypes.c_void_p * 9
    dims = (ctypes.c_int32 * 45)(*([1, 1, 4, 1, 4] * 9))
    buf = torch.zeros(64, device=device)
    st = torch.cuda.current_stream().cuda_stream
    assert raw.glam_pad_group(9, vp(*[buf.data_ptr()] * 9), vp(*[buf.data_ptr()] * 9), dims, 0, st) != 0       # more than 8 tensors
    bad = (ctypes.c_int32 * 5)(1, 4, 4, 3, 4)                                                                  # padded smaller than plain
    assert raw.glam_pad_group(1, (ctypes.c_void_p * 1)(buf.data_ptr()), (ctypes.c_void_p * 1)(buf.data_ptr()), bad, 0, st) != 0
    assert raw.glam_pad_group(0, None, None, None, 0, st) == 0


@pytest.mark.parametrize("D", [60, 45, 92])
def test_pair_pool5_protein_sized_segments_vs_oracle(device, D):
    """dot_and_global_pool5 on ligand x protein sized pairs (410 / 97 / 655 residues), an even and an odd score count (lower
    median), a one-atom ligand, widths that need zero padding (45 -> 48) and two chunks per lane (92): the five statistics and
    their gradients against the oracle's per-pair loop."""
    torch.manual_seed(40 + D)
    nm, npr = [20, 13, 1, 28], [410, 97, 33, 655]
    mol, pro = torch.randn(sum(nm), D), torch.randn(sum(npr), D)
    mb = torch.repeat_interleave(torch.arange(4), torch.tensor(nm))
    pb = torch.repeat_interleave(torch.arange(4), torch.tensor(npr))
    mo, po = mol.clone().requires_grad_(True), pro.clone().requires_grad_(True)
    ref = O.dot_and_global_pool(mo, po, mb, pb, 4, stats=5)
    cot = torch.randn(ref.shape)
    g_ref = _grads(ref, cot, [mo, po])
    m, p = mol.to(device).requires_grad_(True), pro.to(device).requires_grad_(True)
    out = layer.dot_and_global_pool5(m, p, mb.to(device), pb.to(device))

    def run(dt):
        a, b_ = mol.to(dt).requires_grad_(True), pro.to(dt).requires_grad_(True)
        o = O.dot_and_global_pool(a, b_, mb, pb, 4, stats=5)
        return o.detach(), _grads(o, cot.to(dt), [a, b_])
    assert_twin_parity(run, out, _grads(out, cot.to(device), [m, p]), "pool5 stats", ["mol", "pro"])
    out2 = layer.dot_and_global_pool5(m, p, mb.to(device), pb.to(device))
    assert torch.equal(out, out2), "bit-reproducible"
    # the median is an ELEMENT of the score matrix (exact selection, not an interpolation)
    S0 = (mol[:20] @ pro[:410].T).flatten()
    assert (S0 - out[0, 2].cpu()).abs().min().item() <= 1e-5 * max(1.0, S0.abs().max().item())


def test_pair_pool_protein_sized_segments(device):
    torch.manual_seed(4)
    nm, npr, D = [20, 13, 28], [410, 97, 655], 60
    mol, pro = torch.randn(sum(nm), D), torch.randn(sum(npr), D)
    mb = torch.repeat_interleave(torch.arange(3), torch.tensor(nm))
    pb = torch.repeat_interleave(torch.arange(3), torch.tensor(npr))
    mo, po = mol.clone().requires_grad_(True), pro.clone().requires_grad_(True)
    ref = O.dot_and_global_pool(mo, po, mb, pb, 3, stats=2)
    cot = torch.randn(ref.shape)
    g_ref = _grads(ref, cot, [mo, po])
    m, p = mol.to(device).requires_grad_(True), pro.to(device).requires_grad_(True)
    out = layer.dot_and_global_pool2(m, p, mb.to(device), pb.to(device))

    def run(dt):
        a, b_ = mol.to(dt).requires_grad_(True), pro.to(dt).requires_grad_(True)
        o = O.dot_and_global_pool(a, b_, mb, pb, 3, stats=2)
        return o.detach(), _grads(o, cot.to(dt), [a, b_])
    assert_twin_parity(run, out, _grads(out, cot.to(device), [m, p]), "pair pool", ["mol", "pro"])


def test_pair_pool_hands_its_operands_on_with_their_next_gradient_added_inside(device):
    """dot_and_global_pool2(..., with_identity=True) (src_2gi_dti_scr/model.py:66-70: every step's node features feed the fusion AND
    the next message step): the two matrices come back from the fusion node, and the gradient of what the caller does with them next
    is added inside the node's backward launch (glam_pair_pool_bwd_add) — same values, same gradients as the plain call followed by an
    add launch of the autograd engine."""
    from glam_amd._lib import kernel_timer
    torch.manual_seed(4)
    nm, npr, D = [20, 13, 28], [410, 97, 655], 60
    mb = torch.repeat_interleave(torch.arange(3), torch.tensor(nm)).to(device)
    pb = torch.repeat_interleave(torch.arange(3), torch.tensor(npr)).to(device)
    mol, pro = torch.randn(sum(nm), D, device=device), torch.randn(sum(npr), D, device=device)
    wm, wp, cot = torch.randn(sum(nm), D, device=device), torch.randn(sum(npr), D, device=device), torch.randn(3, 2, device=device)
    res = []
    for alias in (True, False):
        m, p = mol.clone().requires_grad_(True), pro.clone().requires_grad_(True)
        mi, pi = m * 1.0, p * 1.0
        if alias:
            out, m2, p2 = layer.dot_and_global_pool2(mi, pi, mb, pb, with_identity=True)
            assert m2.data_ptr() == mi.data_ptr() and p2.data_ptr() == pi.data_ptr()
        else:
            out, m2, p2 = layer.dot_and_global_pool2(mi, pi, mb, pb), mi, pi
        loss = (out * cot).sum() + (m2 * wm).sum() + (p2.square() * wp).sum()
        with kernel_timer() as kt:
            gm, gp = torch.autograd.grad(loss, (m, p))
        res.append((out.detach(), gm, gp, [r[0] for r in kt.records()]))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    assert sum("k_pair_pool_bwd" in n for n in res[0][3]) == 1
    # only one of the two used again / none: still the same
    m, p = mol.clone().requires_grad_(True), pro.clone().requires_grad_(True)
    out, m2, p2 = layer.dot_and_global_pool2(m * 1.0, p * 1.0, mb, pb, with_identity=True)
    gm, gp = torch.autograd.grad((out * cot).sum() + (m2 * wm).sum(), (m, p))
    assert torch.equal(gm, res[1][1]) and not torch.isnan(gp).any()
    lib = ops._lib.load()
    assert lib.glam_pair_pool_add_supported(60) == 1 and lib.glam_pair_pool_add_supported(62) == 0


@pytest.mark.parametrize("kind", ["pair", "layer"])
def test_graph_norms_on_protein_sized_graphs(device, kind):
    """N / B >= 64 routes PairNorm / graph LayerNorm to the block-per-graph kernels."""
    torch.manual_seed(21)
    sizes = [130, 402, 77, 256]
    N, B, D = sum(sizes), len(sizes), 60
    batch = torch.repeat_interleave(torch.arange(B), torch.tensor(sizes))
    x0 = torch.randn(N, D) * 2 + 0.5
    xo = x0.clone().requires_grad_(True)
    w = b_ = None
    if kind == "pair":
        ref = O.pair_norm(xo, batch, B)
        mod = layer.PairNorm().to(device)
    else:
        w, b_ = torch.rand(D) + 0.5, torch.randn(D) * 0.1
        ref = O.graph_layer_norm(xo, w, b_, batch, B)
        mod = layer.LayerNorm(D).to(device)
        with torch.no_grad():
            mod.weight.copy_(w); mod.bias.copy_(b_)
    cot = torch.randn(ref.shape)
    x = x0.to(device).requires_grad_(True)
    out = mod(x, batch.to(device))

    def run(dt):
        xr = x0.to(dt).requires_grad_(True)
        o = O.pair_norm(xr, batch, B) if kind == "pair" else O.graph_layer_norm(xr, w.to(dt), b_.to(dt), batch, B)
        return o.detach(), _grads(o, cot.to(dt), [xr])
    assert_twin_parity(run, out, _grads(out, cot.to(device), [x]), kind, ["x"])


def test_cat_cols_hands_back_contiguous_gradients_from_one_launch(device):
    """ops.cat_cols = torch.cat(dim=-1) (the fusion vector of the two-tower models, src_2gi_dti_scr/model.py:60-75): same value, and in
    the backward every input receives its column block as a CONTIGUOUS tensor from one glam_pad_group launch (autograd's CatBackward
    hands out strided views, copied once per consumer)."""
    from glam_amd._lib import kernel_timer
    torch.manual_seed(2)
    widths = (60, 60, 2, 2, 2, 5, 1)
    ts = [torch.randn(32, c, device=device, requires_grad=(i != 3)) for i, c in enumerate(widths)]
    out = ops.cat_cols(ts)
    assert torch.equal(out, torch.cat(ts, dim=-1)) and type(out.grad_fn).__name__.startswith("_CatCols")
    cot = torch.randn_like(out)
    seen = {}
    for i, t in enumerate(ts):
        if t.requires_grad:
            t.register_hook(lambda g, i=i: seen.__setitem__(i, (g.is_contiguous(), tuple(g.shape))))
    with kernel_timer() as kt:
        out.backward(cot)
    assert [r[0].split("<")[0] for r in kt.records()] == ["k_pad_group"]
    off = 0
    for i, (t, c) in enumerate(zip(ts, widths)):
        if t.requires_grad:
            assert torch.equal(t.grad, cot[:, off:off + c]) and seen[i] == (True, (32, c))
        else:
            assert t.grad is None
        off += c
    # outside its class: plain torch.cat
    assert ops.cat_cols([torch.randn(3, 2), torch.randn(3, 4)]).shape == (3, 6)
    with torch.no_grad():
        assert ops.cat_cols(ts).grad_fn is None


@pytest.mark.parametrize("conv", ["_GCNConv", "_NNConv"])
def test_block_skip_connection_through_the_conv_node(device, monkeypatch, conv):
    """MessageBlock around a GCNConv (the protein tower of the two-tower model, src_1gp/layer.py:248-265; no GRU) or an NNConv (the ligand
    tower; GRU behind it): with ops.SKIP_THROUGH_CONV the block input comes back from the conv's first node (``x @ W``, resp. the relation
    sums) as the skip connection's operand and its gradient joins that node's own d_x inside its backward launch (glam_ts_gemm_add,
    glam_edge_wsum_bwd_add) — same outputs bit for bit, same gradients as with the add launch of the autograd engine, also when the
    block input has a second consumer."""
    torch.manual_seed(5)
    b = synth_batch(64, seed=2).to(device)
    blk = layer.MessageBlock(60, 60, 4, norm="_None", dropout="_None()", conv=conv, act="ReLU()").to(device)
    x0 = torch.randn(b.x.size(0), 60, device=device)
    res = {}
    for flag in (True, False):
        monkeypatch.setattr(ops, "SKIP_THROUGH_CONV", flag)
        x = x0.clone().requires_grad_(True)
        xin = x * 1.5                                   # (a non-leaf input, as inside the model)
        y, hh = blk(xin, b.edge_index, b.edge_attr)
        y2, _ = blk(y, b.edge_index, b.edge_attr, hh if conv == "_NNConv" else None)       # applied twice: the first output feeds the second block AND the loss
        blk.zero_grad()
        (y2.square().sum() + y.sum()).backward()
        res[flag] = (y.detach().clone(), y2.detach().clone(), x.grad.clone(), [p.grad.clone() for p in blk.parameters()])
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    assert_close(res[True][2], res[False][2].double(), 2e-6, "d_x")
    for u, v in zip(res[True][3], res[False][3]):
        assert_close(u, v.double(), 2e-6, "parameter gradient")


@pytest.mark.parametrize("D", [30, 32, 60])
def test_gcn_conv_against_oracle(device, D):
    """GCNConv: cached symmetric normalisation + the K = 1 gather-scale-sum kernels (float4 when D % 4 == 0)."""
    torch.manual_seed(33)
    b = synth_batch(6, seed=4)
    conv = layer.GCNConv(D, D)
    with torch.no_grad():
        conv.bias.uniform_(-0.1, 0.1)
    x0 = torch.randn(b.x.size(0), D)
    w0, b0 = conv.weight.detach().clone(), conv.bias.detach().clone()
    cot = torch.randn(b.x.size(0), D)

    def run(dt):
        xo, w, bias = x0.to(dt).requires_grad_(True), w0.to(dt).requires_grad_(True), b0.to(dt).requires_grad_(True)
        o = O.gcn_conv(xo, b.edge_index, w, bias)
        return o, _grads(o, cot.to(dt), [xo, w, bias])
    conv = conv.to(device)
    x = x0.to(device).requires_grad_(True)
    ei = b.edge_index.to(device)
    for _ in range(2):                   # second pass: normalisation served from the edge-list cache
        out = conv(x, ei)
        assert_twin_parity(run, out, _grads(out, cot.to(device), [x, conv.weight, conv.bias]), "gcn", ["x", "weight", "bias"])


def test_triplet_four_heads_wide_fallback(device):
    """heads = 4 at C = 90 (H*Cp + 8 = 376) is beyond the one-node wide path: torch-sequenced fallback (library GEMMs for the
    data-side products, k_wgrad weight gradients through ops.matmul_tall) around the same aggregate kernels."""
    torch.manual_seed(77)
    b = synth_batch(48, seed=3)
    C, H = 90, 4
    assert not ops.wide_layer_supported(C, H, 4) and ops.wide_layer_supported(C, 3, 4)
    x0 = torch.randn(b.x.size(0), C)
    conv = layer.TripletMessage(C, 4, heads=H)
    ps0 = [p.detach().clone() for p in conv.parameters()]
    cot = torch.randn(b.x.size(0), C)

    def run(dt):
        xo = x0.to(dt).requires_grad_(True)
        ps = [p.to(dt).requires_grad_(True) for p in ps0]
        o = O.triplet_message(xo, b.edge_index, b.edge_attr.to(dt), *ps, heads=H)
        return o, _grads(o, cot.to(dt), [xo] + ps)
    conv = conv.to(device)
    x = x0.to(device).requires_grad_(True)
    out = conv(x, b.edge_index.to(device), b.edge_attr.to(device))
    assert_twin_parity(run, out, _grads(out, cot.to(device), [x] + list(conv.parameters())), "four heads, C = 90",
                       ["x"] + [n for n, _ in conv.named_parameters()])


@pytest.mark.parametrize("C,H", [(30, 6), (60, 5), (45, 8)])
def test_triplet_message_more_than_four_heads(device, C, H):
    """``TripletMessage(heads > 4)`` (layer.py:16 accepts any head count; the kernels hold four per lane group): the layer as a sum over
    head groups on parameter slices — output and every gradient against the oracle's direct formulation."""
    torch.manual_seed(C + H)
    b = synth_batch(16, seed=H)
    x0 = torch.randn(b.x.size(0), C)
    conv = layer.TripletMessage(C, 4, heads=H)
    ps0 = [p.detach().clone() for p in conv.parameters()]
    cot = torch.randn(b.x.size(0), C)

    def run(dt):
        xo = x0.to(dt).requires_grad_(True)
        ps = [p.to(dt).requires_grad_(True) for p in ps0]
        o = O.triplet_message(xo, b.edge_index, b.edge_attr.to(dt), *ps, heads=H)
        return o, _grads(o, cot.to(dt), [xo] + ps)
    conv = conv.to(device)
    x = x0.to(device).requires_grad_(True)
    out = conv(x, b.edge_index.to(device), b.edge_attr.to(device))
    assert_twin_parity(run, out, _grads(out, cot.to(device), [x] + list(conv.parameters())), f"{H} heads, C = {C}",
                       ["x"] + [n for n, _ in conv.named_parameters()])


@pytest.mark.parametrize("alpha,act,block", [(1, "ReLU", "_TripletMessage"), (2, "ReLU", "_TripletMessage"), (3, "CELU", "_TripletMessage"),
                                             (6, "ReLU", "_TripletMessage"), (3, "ReLU", "_NNConv"), (2, "LeakyReLU", "_TripletMessageLight")])
def test_architecture_odd_widths_vs_oracle(device, alpha, act, block):
    """hid_dim_alpha of the search space (glam.py:60) whose hidden width is not a multiple of 4: features travel between the
    conv and the GRU step as zero-padded rows handed on by reference (ops.pad_cols / slice_cols); outputs and every parameter
    gradient against the oracle's full model."""
    torch.manual_seed(40 + alpha)
    b = synth_batch(24, seed=alpha)
    net = model.Architecture(hid_dim_alpha=alpha, e_dim=64, out_dim=2, message_steps=3, mol_block=block, mol_readout="GlobalPool5",
                             graph_norm="_None", pre_act=act, graph_act=act, flat_act=act, graph_do="_None()", end_do="_None()").eval()
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    ref = O.architecture(sd, b, b.num_graphs, message_steps=3, mol_block=block, mol_readout="GlobalPool5", graph_norm="_None",
                         pre_act=act, graph_act=act, flat_act=act)
    cot = torch.randn(ref.shape)
    names = [n for n, _ in net.named_parameters()]
    g_ref = _grads(ref, cot, [sd[n] for n in names])
    net_cpu_sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.to(device)
    out = net(b.to(device))

    def run(dt):
        sd_ = {k: v.to(dt).clone().requires_grad_(True) for k, v in net_cpu_sd.items()}
        bb = type(b)(b.x.to(dt), b.edge_index, b.edge_attr.to(dt), batch=b.batch)
        o = O.architecture(sd_, bb, b.num_graphs, message_steps=3, mol_block=block, mol_readout="GlobalPool5", graph_norm="_None",
                           pre_act=act, graph_act=act, flat_act=act)
        return o.detach(), _grads(o, cot.to(dt), [sd_[n] for n in names])
    assert_twin_parity(run, out, _grads(out, cot.to(device), [p for _, p in net.named_parameters()]), "odd width", names)


def test_padded_views_are_not_trusted_after_inplace_writes(device):
    """ops.pad_cols only reuses a registered padded tensor while nobody wrote to it (version counter)."""
    xp = torch.zeros(8, 48, device=device)
    xp[:, :45] = torch.randn(8, 45, device=device)
    v = ops.slice_cols(xp, 45)
    assert ops.pad_cols(v, 48) is xp
    v.add_(1.0)                                             # pad columns may no longer be what the producer left
    p = ops.pad_cols(v, 48)
    assert p is not xp and torch.equal(p[:, :45], v) and (p[:, 45:] == 0).all()
    w = torch.randn(8, 48, device=device)[:, :45]           # a view nobody registered
    q = ops.pad_cols(w, 48)
    assert q.data_ptr() != w.data_ptr() and (q[:, 45:] == 0).all()


@pytest.mark.parametrize("kind", ["pair", "layer"])
@pytest.mark.parametrize("D", [60, 32])
def test_graph_norms_mixed_graph_sizes(device, kind, D):
    """Wave-per-graph norm kernels at padded widths: graphs below and above the 32-node register pass, single-node graphs."""
    torch.manual_seed(23)
    sizes = [40, 3, 33, 50, 1, 32, 17]
    N, B = sum(sizes), len(sizes)
    batch = torch.repeat_interleave(torch.arange(B), torch.tensor(sizes))
    x0 = torch.randn(N, D) * 1.5 - 0.3
    xo = x0.clone().requires_grad_(True)
    w = b_ = None
    if kind == "pair":
        ref = O.pair_norm(xo, batch, B)
        mod = layer.PairNorm().to(device)
    else:
        w, b_ = torch.rand(D) + 0.5, torch.randn(D) * 0.1
        ref = O.graph_layer_norm(xo, w, b_, batch, B)
        mod = layer.LayerNorm(D).to(device)
        with torch.no_grad():
            mod.weight.copy_(w); mod.bias.copy_(b_)
    cot = torch.randn(ref.shape)
    x = x0.to(device).requires_grad_(True)
    out = mod(x, batch.to(device))

    def run(dt):
        xr = x0.to(dt).requires_grad_(True)
        o = O.pair_norm(xr, batch, B) if kind == "pair" else O.graph_layer_norm(xr, w.to(dt), b_.to(dt), batch, B)
        return o.detach(), _grads(o, cot.to(dt), [xr])
    assert_twin_parity(run, out, _grads(out, cot.to(device), [x]), kind, ["x"])


def test_wgrad_gemm_chunked_with_ones_column(device, wgrad_route):
    """[d_W | d_b] = dy^T [x | 1] with 64 < K + 1 <= 128: the ones column rides on the second column chunk."""
    from glam_amd import _lib
    lib, p = _lib.load(), _lib.ptr
    g = torch.Generator().manual_seed(5)
    N, M, K = 6000, 276, 92
    dy, x = torch.randn(N, M, generator=g), torch.randn(N, K, generator=g)
    ref = dy.double().t() @ torch.cat([x, torch.ones(N, 1)], 1).double()
    ws = torch.empty(lib.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=device)
    out = torch.full((M, K + 1), float("nan"), device=device)
    dyd, xd = dy.to(device), x.to(device)
    rc = lib.glam_wgrad_gemm(p(dyd), M, M, None, 0, 0, 0, p(xd), K, K, 1, N, p(out), K + 1, 1, p(ws), ws.numel(), _lib.stream())
    assert rc == 0, lib.glam_last_error()
    assert_close(out, ref, 3e-6 * N ** 0.5 / 10, "chunked wgrad + ones")


@pytest.mark.parametrize("C,sizes", [(60, [20, 13, 1, 28]), (45, [7, 40, 33, 2]), (32, [70, 5]), (90, [19, 3, 40])])
def test_set2set_fused_steps_vs_oracle(device, C, sizes):
    """Set2Set on the fused path (LSTM gate kernel + in-kernel query attention): graphs below and above the 32-node register
    pass, an odd width (padded rows) — output and every gradient against the oracle."""
    torch.manual_seed(50 + C)
    N, B = sum(sizes), len(sizes)
    batch = torch.repeat_interleave(torch.arange(B), torch.tensor(sizes))
    ro = layer.Set2Set(C, 3)
    x0 = torch.randn(N, C)
    xo = x0.clone().requires_grad_(True)
    import copy
    lstm_ref = copy.deepcopy(ro.lstm)
    names = [n for n, _ in lstm_ref.named_parameters()]
    ref = O.set2set(xo, batch, B, lstm_ref, steps=3)
    cot = torch.randn(ref.shape)
    g_ref = _grads(ref, cot, [xo] + [p for _, p in lstm_ref.named_parameters()])
    ro = ro.to(device)
    x = x0.to(device).requires_grad_(True)
    out = ro(x, batch.to(device), B)

    def run(dt):
        l2 = copy.deepcopy(lstm_ref).to(dt)
        xr = x0.to(dt).requires_grad_(True)
        o = O.set2set(xr, batch, B, l2, steps=3)
        return o.detach(), _grads(o, cot.to(dt), [xr] + [p for _, p in l2.named_parameters()])
    assert_twin_parity(run, out, _grads(out, cot.to(device), [x] + [p for _, p in ro.lstm.named_parameters()]), "set2set", ["x"] + names)


def test_misuse_raises_python_exceptions_and_leaves_the_device_usable(device):
    """Error behaviour at the boundary (the reference convention is Python exceptions, never an abort)."""
    b = synth_batch(8, seed=0).to(device)
    conv = layer.TripletMessage(60, 4).to(device)
    x = torch.randn(b.x.size(0), 60, device=device)
    ok = conv(x, b.edge_index, b.edge_attr)
    with pytest.raises(ops.GlamHipError):
        conv(x, b.edge_index.int(), b.edge_attr)                      # int64 edge_index only
    with pytest.raises(ops.GlamHipError):
        conv(x.double(), b.edge_index, b.edge_attr)                   # fp32 only
    with pytest.raises(ops.GlamHipError):
        conv(x, b.edge_index, b.edge_attr[:-3])                       # one edge_attr row per edge
    with pytest.raises(IndexError):
        conv(x[:-2], b.edge_index, b.edge_attr)                       # node ids beyond x
    ei = b.edge_index.clone()
    ei[1, 0] = -1
    with pytest.raises(IndexError):
        conv(x, ei, b.edge_attr)
    with pytest.raises(ops.GlamHipError):
        conv(x.cpu(), b.edge_index, b.edge_attr)                      # no CPU fallback
    with pytest.raises(ops.GlamHipError):
        layer.TripletMessage(60, 9).to(device)(x, b.edge_index, torch.rand(b.edge_index.size(1), 9, device=device))
    with pytest.raises(IndexError):
        layer.GlobalPool5()(x, b.batch.flip(0))                       # batch must be sorted (it is, by collation)
    # negative / huge graph ids anywhere in `batch`: IndexError, and no write outside the ptr buffer (k_batch_ptr used to run its
    # fill loop from prev + 1 < 0 on the element AFTER a negative id)
    canary = torch.zeros(4096, dtype=torch.int32, device=device)
    for pos, val in ((0, -5), (3, -(2 ** 40)), (b.batch.numel() - 1, -1), (b.batch.numel() - 1, 2 ** 40), (2, 2 ** 33)):
        bad = b.batch.clone()
        bad[pos] = val
        with pytest.raises(IndexError):
            ops.SegmentPtr(bad, 8 if pos != b.batch.numel() - 1 else None)
    torch.cuda.synchronize()
    assert int(canary.abs().sum()) == 0
    xt = torch.randn(60, b.x.size(0), device=device).t()              # non-contiguous input: accepted
    assert conv(xt, b.edge_index, b.edge_attr).shape == ok.shape
    assert torch.equal(conv(x, b.edge_index, b.edge_attr), ok)        # the device is still fine


def test_gradient_carry_is_bit_identical_to_autograd_accumulation(device, monkeypatch):
    """Inside a weight scope the shared block's parameter gradients travel from application to application as one flat
    buffer (ops._ParamBundle); same sums in the same order as autograd's per-tensor accumulation."""
    torch.manual_seed(9)
    b = synth_batch(48, seed=2).to(device)
    net = model.Architecture(message_steps=3, mol_block="_TripletMessage", graph_norm="_None", graph_do="_None()", end_do="_None()",
                             pre_act="ReLU", graph_act="ReLU", flat_act="ReLU").to(device).eval()
    params = [p for _, p in net.named_parameters()]
    grads = {}
    monkeypatch.setattr(ops, "USE_TORCH_EXT", False)        # the carry is the Python node's (what a captured step runs)
    # (one product per application + the carried addend: the form this test pins; the default — ONE product over the parked operand sets
    #  of all applications, test_gru_weight_gradients_of_all_applications_in_one_launch — sums in another order)
    monkeypatch.setattr(ops, "GRU_WGRAD_BATCH", False)
    for flag in (True, False):
        ops.GRAD_CARRY = flag
        try:
            out = net(b)
            grads[flag] = torch.autograd.grad(out.sum(), params)
        finally:
            ops.GRAD_CARRY = True
    for (n, _), a, r in zip(net.named_parameters(), grads[True], grads[False]):
        assert torch.equal(a, r), n
    # .backward(): the carried gradients land in .grad as views of one buffer per parameter set (no copies)
    net.zero_grad(set_to_none=True)
    net(b).sum().backward()
    conv = net.mol_conv.conv.conv
    ptrs = [p.grad.data_ptr() for p in (conv.weight_node, conv.weight_edge, conv.weight_triplet_att, conv.weight_scale, conv.bias)]
    assert ptrs == sorted(ptrs) and ptrs[-1] - ptrs[0] < 4 * sum(p.numel() for p in conv.parameters())
    for (n, p), r in zip(net.named_parameters(), grads[False]):
        assert torch.equal(p.grad, r), n
    # an eagerly issued step takes the torch-extension operator (no carry: autograd accumulates): same bits again
    monkeypatch.setattr(ops, "USE_TORCH_EXT", "auto")
    for (n, _), a, r in zip(net.named_parameters(), torch.autograd.grad(net(b).sum(), params), grads[False]):
        assert torch.equal(a, r), n


@pytest.mark.parametrize("block,norm", [("_NNConv", "_None"), ("_NNConv", "_PairNorm"), ("_GCNConv", "_None")])
def test_gradient_carry_of_tall_matmul_weights(device, block, norm, monkeypatch):
    """NNConv's stacked relation weight + bias (and GCN's weight) enter ``ops.matmul_tall`` once per message step: inside a weight
    scope their gradients are carried through the weight-gradient reductions (glam_wgrad_gemm_add) instead of being summed by
    autograd's add launches — the same sums in the same order, so the same bits."""
    torch.manual_seed(11)
    b = synth_batch(48, seed=4).to(device)
    net = model.Architecture(message_steps=3, mol_block=block, graph_norm=norm, graph_do="_None()", end_do="_None()",
                             pre_act="ReLU", graph_act="ReLU", flat_act="ReLU").to(device).eval()
    params = [p for _, p in net.named_parameters()]
    grads = {}
    monkeypatch.setattr(ops, "GRU_WGRAD_BATCH", False)      # (the per-application form: see the test above)
    for flag in (True, False):
        ops.GRAD_CARRY = flag
        try:
            grads[flag] = torch.autograd.grad(net(b).sum(), params)
        finally:
            ops.GRAD_CARRY = True
    for (n, _), a, r in zip(net.named_parameters(), grads[True], grads[False]):
        assert torch.equal(a, r), n
    monkeypatch.setattr(ops, "GRU_WGRAD_BATCH", True)       # one product over all applications: the same sums in another order
    for (n, _), a, r in zip(net.named_parameters(), torch.autograd.grad(net(b).sum(), params), grads[False]):
        assert_close(a, r, 3e-6, n)


@pytest.mark.parametrize("kind", ["pair", "layer"])
@pytest.mark.parametrize("mols,width", [(48, 60), (7, 45), (3, 64)])
def test_graph_norm_with_identity_sums_both_gradient_paths(device, kind, mols, width):
    """``ops.pair_norm / graph_standardize(..., with_identity=True)`` hand ``x`` back as a second output (the skip connection of a
    MessageBlock, src_1gp/layer.py:253-265); the backward kernel adds that path's gradient in its store (glam_graph_norm_bwd_add):
    the same bits as autograd's add of the two paths."""
    torch.manual_seed(mols + width)
    b = synth_batch(mols, seed=5).to(device)
    sp = ops.segment_ptr(b.batch)
    N = b.x.size(0)
    x0 = torch.randn(N, width, device=device)
    c1, c2 = torch.randn(N, width, device=device), torch.randn(N, width, device=device)
    f = ops.pair_norm if kind == "pair" else ops.graph_standardize
    xa = x0.clone().requires_grad_(True)
    ya, ida = f(xa, sp, with_identity=True)
    assert ida.data_ptr() == xa.data_ptr()
    (ga,) = torch.autograd.grad((ya * c1).sum() + (ida * c2).sum(), [xa])
    xb = x0.clone().requires_grad_(True)
    yb = f(xb, sp)
    (gb,) = torch.autograd.grad((yb * c1).sum() + (xb * c2).sum(), [xb])
    assert torch.equal(ya, yb) and torch.equal(ga, gb)
    # only one of the two outputs used
    xc = x0.clone().requires_grad_(True)
    yc, idc = f(xc, sp, with_identity=True)
    assert torch.equal(torch.autograd.grad((idc * c2).sum(), [xc])[0], c2)
    xd = x0.clone().requires_grad_(True)
    yd, _ = f(xd, sp, with_identity=True)
    assert torch.equal(torch.autograd.grad((yd * c1).sum(), [xd])[0], torch.autograd.grad((f(xb, sp) * c1).sum(), [xb])[0])


def test_sharded_gradients_sum_to_the_single_device_gradient(device):
    """the correctness oracle of the data-parallel path is the single-device run on the concatenated batch.
    Two node-balanced graph shards through DataParallelStep (one process: the all-reduce is the identity) summed by hand
    against one backward on the full batch, on the HIP path."""
    from glam_amd.parallel import DataParallelStep, shard_batch
    torch.manual_seed(0)
    net = model.Architecture(message_steps=3, mol_block="_TripletMessage", graph_norm="_None", graph_do="_None()", end_do="_None()",
                             pre_act="ReLU", graph_act="ReLU", flat_act="ReLU").to(device).eval()
    full = synth_batch(64, seed=5)
    mse = torch.nn.functional.mse_loss
    net.zero_grad(set_to_none=True)
    b = full.to(device)
    mse(net(b).view(-1), b.y.view(-1)).backward()
    ref = [p.grad.clone() for p in net.parameters()]
    step = DataParallelStep(net, lambda out, s: (mse(out.view(-1), s.y.view(-1), reduction="sum") / 64, 1.0))
    tot = None
    for r in range(2):
        step(shard_batch(full, r, 2).to(device))
        g = step.bucket.flat.clone()
        tot = g if tot is None else tot + g
    twin = {}
    for dt in (torch.float32, torch.float64):      # the oracle on the concatenated batch, both precisions
        sd = {k: v.detach().cpu().to(dt).requires_grad_(True) for k, v in net.state_dict().items()}
        data = type(full)(full.x.to(dt), full.edge_index, full.edge_attr.to(dt), batch=full.batch)
        o = O.architecture(sd, data, 64, message_steps=3, mol_block="_TripletMessage", mol_readout="GlobalPool5", pre_act="ReLU",
                           graph_act="ReLU", flat_act="ReLU")
        twin[dt] = torch.autograd.grad(mse(o.view(-1), full.y.view(-1).to(dt)), [sd[n] for n, _ in net.named_parameters()])
    off = 0
    for (n, p), r_, r64, r32 in zip(net.named_parameters(), ref, twin[torch.float64], twin[torch.float32]):
        assert_fp32_parity(tot[off:off + p.numel()].view_as(p), r64, r32, "sharded grad " + n)
        assert_fp32_parity(r_, r64, r32, "single-device grad " + n)
        off += p.numel()


def test_readouts_and_norms_with_empty_and_tiny_graphs(device):
    """Graph ids that skip values (empty graphs), single-node graphs and graphs with fewer than k = 3 nodes through the
    wave-per-graph kernels at a padded width: GlobalPool5, PairNorm, Set2Set and GlobalLAPool against the oracle."""
    torch.manual_seed(61)
    C = 60
    sizes = {0: 5, 2: 1, 3: 2, 5: 34, 6: 3}          # graphs 1 and 4 are empty; B = 8 with a trailing empty graph
    B = 8
    batch = torch.cat([torch.full((n,), g) for g, n in sizes.items()])
    N = batch.numel()
    x0 = torch.randn(N, C)
    bd = batch.to(device)

    def compare(ref_fn, dev_fn, what):
        """``ref_fn(x, dtype)``: the oracle in the given precision (module-backed oracles get a copy of their module in it)."""
        x = x0.to(device).requires_grad_(True)
        out = dev_fn(x)
        cot = torch.randn(out.shape)

        def run(dt):
            xr = x0.to(dt).requires_grad_(True)
            o = ref_fn(xr, dt)
            return o.detach(), _grads(o, cot.to(dt), [xr])
        assert_twin_parity(run, out, _grads(out, cot.to(device), [x]), what, ["x"])

    compare(lambda x, dt: O.global_pool5(x, batch, B), lambda x: layer.GlobalPool5()(x, bd, B), "pool5")
    compare(lambda x, dt: O.pair_norm(x, batch, B), lambda x: ops.pair_norm(x, ops.segment_ptr(bd, B)), "pair_norm")
    ro = layer.Set2Set(C, 3)
    import copy
    lstm_ref = copy.deepcopy(ro.lstm)
    ro = ro.to(device)
    compare(lambda x, dt: O.set2set(x, batch, B, copy.deepcopy(lstm_ref).to(dt), steps=3), lambda x: ro(x, bd, B), "set2set")
    la = layer.GlobalLAPool(C)
    sd = {k: v.detach().clone() for k, v in la.state_dict().items()}
    la = la.to(device)
    compare(lambda x, dt: O.global_attention(x, batch, B, sd["pool.gate_nn.weight"].to(dt), sd["pool.gate_nn.bias"].to(dt),
                                             sd["pool.nn.weight"].to(dt), sd["pool.nn.bias"].to(dt)),
            lambda x: la(x, bd, B), "lapool")


@pytest.mark.parametrize("C", [45, 60, 90])
def test_empty_batch_forward_and_backward(device, C):
    """N = 0 (an empty shard of a data-parallel step) on every width class: outputs are empty, parameter gradients zero."""
    conv = layer.TripletMessage(C, 4).to(device)
    x = torch.zeros(0, C, device=device, requires_grad=True)
    out = conv(x, torch.zeros(2, 0, dtype=torch.long, device=device), torch.zeros(0, 4, device=device))
    assert out.shape == (0, C)
    gs = torch.autograd.grad(out.sum(), [x] + list(conv.parameters()), allow_unused=True)
    for g_ in gs[1:]:
        assert g_ is None or float(g_.abs().max()) == 0.0


@pytest.mark.parametrize("block,readout", [("_TripletMessage", "GlobalPool5"), ("_NNConv", "Set2Set"), ("_GCNConv", "GlobalLAPool"),
                                           ("_TripletMessageLight", "GlobalPool5"), ("_GATConv", "GlobalPool5")])
def test_full_model_on_an_empty_shard(device, block, readout):
    """More ranks than graphs: a data-parallel rank may own nothing.  The whole model runs on the empty shard — output
    [0, out_dim], every parameter gradient exactly zero — so that the gradient all-reduce stays collective."""
    from glam_amd.parallel import shard_batch
    full = synth_batch(4, seed=0)
    shard = next(sh for sh in (shard_batch(full, r, 8) for r in range(8)) if sh.num_graphs == 0).to(device)
    net = model.Architecture(mol_block=block, mol_readout=readout, graph_norm="_PairNorm", graph_do="_None()", end_do="_None()",
                             pre_act="ReLU", graph_act="ReLU", flat_act="ReLU").to(device)
    out = net(shard)
    assert out.shape == (0, 1)
    out.sum().backward()
    assert all(float(p.grad.abs().max()) == 0.0 for p in net.parameters() if p.grad is not None)


def test_graph_res_zero_keeps_the_residual_like_the_reference(device):
    """src_1gp/layer.py:265 tests ``self.res is False``: ``graph_res=0`` (the int that run.py:38 / glam.py:81 pass) is not
    ``False``, so the reference still adds the residual; only the literal ``False`` removes it.  Mirrored, not "fixed"."""
    torch.manual_seed(3)
    b = synth_batch(6, seed=1).to(device)
    x = torch.randn(b.x.size(0), 60, device=device)
    outs = []
    for res in (True, 1, 0, False):                    # (a dict would merge the keys 0 / False and 1 / True)
        torch.manual_seed(11)
        blk = layer.MessageBlock(60, 60, 4, norm="_None", dropout="_None()", conv="_TripletMessage", act="ReLU", res=res).to(device).eval()
        outs.append(blk(x, b.edge_index, b.edge_attr, batch=b.batch)[0])
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert not torch.equal(outs[0], outs[3])


@pytest.mark.parametrize("mol_block,pro_block,norm", [("_NNConv", "_GCNConv", "_None"), ("_TripletMessage", "_TripletMessage", "_PairNorm"),
                                                     ("_NNConv", "_GATConv", "_LayerNorm")])
def test_two_tower_model_vs_oracle(device, mol_block, pro_block, norm):
    """ArchitectureDTI (BASELINE config 5) against the oracle's restatement of src_2gi_dti_scr/model.py:45-68: output and
    every parameter gradient, protein graphs large enough for the block-per-graph kernels."""
    torch.manual_seed(12)
    mb = synth_batch(5, seed=3)
    pb = synth_protein_batch(5, seed=4, n_min=40, n_max=130)
    net = model.ArchitectureDTI(mol_block=mol_block, pro_block=pro_block, e_dim=64, message_steps=2, graph_norm=norm,
                                pre_act="ReLU", graph_act="ReLU", flat_act="ReLU", end_act="ReLU", graph_do="_None()", end_do="_None()").eval()
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    ref = O.architecture_dti(sd, mb, pb, 5, message_steps=2, mol_block=mol_block, pro_block=pro_block, graph_norm=norm,
                             pre_act="ReLU", graph_act="ReLU", flat_act="ReLU", end_act="ReLU")
    cot = torch.randn(ref.shape)
    names = [n for n, _ in net.named_parameters()]
    g_ref = torch.autograd.grad((ref * cot).sum(), [sd[n] for n in names], allow_unused=True)
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.to(device)
    out = net(mb.to(device), pb.to(device))
    gs = torch.autograd.grad((out * cot.to(device)).sum(), [p for _, p in net.named_parameters()], allow_unused=True)

    def run(dt):
        sd_ = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd0.items()}
        cast = lambda b: type(b)(b.x.to(dt), b.edge_index, b.edge_attr.to(dt), batch=b.batch)
        o = O.architecture_dti(sd_, cast(mb), cast(pb), 5, message_steps=2, mol_block=mol_block, pro_block=pro_block, graph_norm=norm,
                               pre_act="ReLU", graph_act="ReLU", flat_act="ReLU", end_act="ReLU")
        return o.detach(), torch.autograd.grad((o * cot.to(dt)).sum(), [sd_[n] for n in names], allow_unused=True)
    assert_twin_parity(run, out, gs, "dti", names)


@pytest.mark.parametrize("name", ["ddi_nnconv", "ddi_triplet"])
def test_two_drug_architecture_golden(device, name):
    """ArchitectureDDI (src_2gi_ddi/model.py:9-62: two ligand towers + per-pair fusion) on the HIP path against the vectors captured from
    the reference's own model (oracle/gen_goldens.py); bounds from the oracle's fp64 run on the same inputs.  Also: the state dict
    keys and a seeded construction equal the fixture's (the reference's parameter creation order)."""
    g = Golden(name)
    m = g.meta
    kw = dict(e_dim=m["e_dim"], message_steps=m["message_steps"], hid_dim_alpha=m["hid_dim_alpha"], mol_block=m["mol_block"],
              graph_norm=m["graph_norm"], pre_act=m["pre_act"], graph_act=m["graph_act"], flat_act=m["flat_act"], end_act=m["end_act"],
              graph_do="_None()", end_do="_None()")
    net = model.ArchitectureDDI(**kw)
    assert list(net.state_dict().keys()) == list(g.params.keys())
    net.load_state_dict(g.params)
    net = net.to(device).eval()
    i = _dev(g.inputs, device)
    m1 = Data(i["mol1_x"], i["mol1_edge_index"], i["mol1_edge_attr"], batch=i["mol1_batch"])
    m2 = Data(i["mol2_x"], i["mol2_edge_index"], i["mol2_edge_attr"], batch=i["mol2_batch"])
    m1.num_graphs = m2.num_graphs = m["B"]
    out = net(m1, m2)
    names = [n for n, _ in net.named_parameters()]
    sd64 = {k: v.double().clone().requires_grad_(True) for k, v in g.params.items()}
    ii = g.inputs
    a64 = Data(ii["mol1_x"].double(), ii["mol1_edge_index"], ii["mol1_edge_attr"].double(), batch=ii["mol1_batch"])
    b64 = Data(ii["mol2_x"].double(), ii["mol2_edge_index"], ii["mol2_edge_attr"].double(), batch=ii["mol2_batch"])
    o64 = O.architecture_ddi(sd64, a64, b64, m["B"], message_steps=m["message_steps"], mol_block=m["mol_block"], graph_norm=m["graph_norm"],
                             pre_act=m["pre_act"], graph_act=m["graph_act"], flat_act=m["flat_act"], end_act=m["end_act"])
    g64 = torch.autograd.grad((o64 * g.cot.double()).sum(), [sd64[n] for n in names], allow_unused=True)
    assert_fp32_parity(out, o64.detach(), g.out, f"{name} out (golden)", out_tol=1e-5)
    for n, t, r64 in zip(names, _grads(out, g.cot.to(device), [p for _, p in net.named_parameters()]), g64):
        if r64 is not None:
            assert_fp32_parity(t, r64, g.grads[n], f"{name}/grad.{n} (golden)")


def test_two_tower_architecture_golden(device):
    """ArchitectureDTI on the HIP path against the vectors captured from the reference's own two-tower model
    (src_2gi_dti_scr/model.py run over the PyG stand-in, oracle/gen_goldens.py)."""
    g = Golden("dti_nnconv_gcn")
    m = g.meta
    net = model.ArchitectureDTI(e_dim=m["e_dim"], message_steps=m["message_steps"], mol_block=m["mol_block"], pro_block=m["pro_block"],
                                graph_norm=m["graph_norm"], pre_act=m["pre_act"], graph_act=m["graph_act"], flat_act=m["flat_act"],
                                end_act=m["end_act"], graph_do="_None()", end_do="_None()")
    net.load_state_dict(g.params)
    net = net.to(device).eval()
    i = _dev(g.inputs, device)
    mol = Data(i["mol_x"], i["mol_edge_index"], i["mol_edge_attr"], batch=i["mol_batch"])
    pro = Data(i["pro_x"], i["pro_edge_index"], i["pro_edge_attr"], batch=i["pro_batch"])
    mol.num_graphs = pro.num_graphs = m["B"]
    out = net(mol, pro)
    names = [n for n, _ in net.named_parameters()]
    # the golden IS the reference's fp32 sample; its rounding-noise floor comes from the oracle's fp64 run on the same inputs
    sd64 = {k: v.double().clone().requires_grad_(True) for k, v in g.params.items()}
    ii = g.inputs
    m64 = Data(ii["mol_x"].double(), ii["mol_edge_index"], ii["mol_edge_attr"].double(), batch=ii["mol_batch"])
    p64 = Data(ii["pro_x"].double(), ii["pro_edge_index"], ii["pro_edge_attr"].double(), batch=ii["pro_batch"])
    o64 = O.architecture_dti(sd64, m64, p64, m["B"], message_steps=m["message_steps"], mol_block=m["mol_block"], pro_block=m["pro_block"],
                             graph_norm=m["graph_norm"], pre_act=m["pre_act"], graph_act=m["graph_act"], flat_act=m["flat_act"],
                             end_act=m["end_act"])
    g64 = torch.autograd.grad((o64 * g.cot.double()).sum(), [sd64[n] for n in names], allow_unused=True)
    assert_fp32_parity(out, o64.detach(), g.out, "dti out (golden)", out_tol=1e-5)
    for n, t, r64 in zip(names, _grads(out, g.cot.to(device), [p for _, p in net.named_parameters()]), g64):
        if r64 is not None:
            assert_fp32_parity(t, r64, g.grads[n], f"dti/grad.{n} (golden)")


def test_graph_index_does_not_pin_edge_index_and_survives_its_death(device):
    """The cached CSR must not keep ``edge_index`` (and itself) alive — a loader that builds new batches every step would
    leak one staging per step — and a backward whose ``edge_index`` was dropped after the forward still gets its transpose."""
    import gc, weakref
    b = synth_batch(5, seed=8).to(device)
    conv = layer.TripletMessage(60, 4).to(device)
    x = torch.randn(b.x.size(0), 60, device=device, requires_grad=True)
    ref_out = conv(x, b.edge_index, b.edge_attr)
    (ref_g,) = torch.autograd.grad(ref_out.sum(), [x])
    ei = b.edge_index.clone()
    wr = weakref.ref(ei)
    gc.collect()                                                      # entries of earlier tests whose tensors sit in cycles
    n_before = len(ops._GI_CACHE)
    out = conv(x, ei, b.edge_attr)
    assert len(ops._GI_CACHE) == n_before + 1
    del ei
    gc.collect()
    assert wr() is None and len(ops._GI_CACHE) == n_before          # tensor and cache entry are gone
    (g,) = torch.autograd.grad(out.sum(), [x])                        # transpose rebuilt from the by-target CSR
    assert torch.equal(out, ref_out) and torch.equal(g, ref_g)


def test_side_stream_gives_identical_results(device):
    """Every entry point enqueues on torch's current stream: the same step on a side stream, bit for bit."""
    torch.manual_seed(0)
    net = model.Architecture(graph_norm="_PairNorm", graph_do="_None()", end_do="_None()", pre_act="ReLU", graph_act="ReLU",
                             flat_act="ReLU").to(device)
    b = synth_batch(48, seed=1).to(device)

    def run():
        net.zero_grad(set_to_none=True)
        out = net(b)
        out.sum().backward()
        return out.detach().clone(), [p.grad.clone() for p in net.parameters()]

    ref_out, ref_g = run()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out, g = run()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    assert torch.equal(out, ref_out) and all(torch.equal(a, r) for a, r in zip(g, ref_g))


# ---------------------------------------------------------------------------------------------
# MessagePassing surface (.propagate / .message / .update are part of the contract)
# ---------------------------------------------------------------------------------------------
def test_message_passing_propagate_message_update_surface(device):
    """propagate() with the tensors the reference's forward hands it (x @ weight_node, edge_attr @ weight_edge:
    layer.py:37-40) runs PyG's collect -> message -> aggregate -> update pipeline and must equal the fused forward and the
    oracle (output + gradients); message() alone equals the oracle's per-edge messages; a user-defined subclass gets the
    generic pipeline with add / mean / max aggregation."""
    torch.manual_seed(21)
    b = synth_batch(40, seed=21)
    N, E = b.x.size(0), b.edge_index.size(1)
    ei, ea = b.edge_index.to(device), b.edge_attr.to(device)
    for conv, C in ((layer.TripletMessage(60, 4), 60), (layer.TripletMessage(30, 4, heads=2), 30), (layer.TripletMessageLight(45, 4), 45)):
        light = isinstance(conv, layer.TripletMessageLight)
        with torch.no_grad():
            conv.bias.normal_(0, 0.1)
        x0 = torch.randn(N, C)
        ps0 = [p.detach().clone() for p in conv.parameters()]
        heads = 1 if light else conv.heads
        cot = torch.randn(N, C)

        def run(dt, light=light, ps0=ps0, x0=x0, heads=heads, cot=cot):
            xo = x0.to(dt).requires_grad_(True)
            ps = [p.to(dt).requires_grad_(True) for p in ps0]
            o = (O.triplet_message_light(xo, b.edge_index, b.edge_attr.to(dt), *ps) if light else
                 O.triplet_message(xo, b.edge_index, b.edge_attr.to(dt), *ps, heads=heads))
            return o, _grads(o, cot.to(dt), [xo] + ps)
        ref = run(torch.float32)[0].detach()
        conv = conv.to(device)
        x = x0.to(device).requires_grad_(True)
        fused = conv(x, ei, ea)
        xw = x @ conv.weight_node
        ew = ea if light else ea @ conv.weight_edge
        out = conv.propagate(ei, x=xw, edge_attr=ew)                 # the reference's own call form
        assert_close(out, fused, TOL, "propagate vs fused forward")
        assert_twin_parity(run, out, _grads(out, cot.to(device), [x] + list(conv.parameters())), "propagate",
                           ["x"] + [n for n, _ in conv.named_parameters()])
        # message() on hand-lifted tensors == what the pipeline aggregates
        msg = conv.message(x_j=xw[ei[0]], x_i=xw[ei[1]], edge_index_i=ei[1], edge_attr=ew, size_i=N)
        agg = torch.zeros((N,) + tuple(msg.shape[1:]), device=device).index_add_(0, ei[1], msg)
        assert_fp32_parity(conv.update(agg), run(torch.float64)[0].detach(), ref, "message + scatter + update", out_tol=1e-5)

    class EdgeGated(layer.MessagePassing):                           # a PyG-style user subclass
        def __init__(self, aggr):
            super().__init__(aggr=aggr)

        def forward(self, x, edge_index, gate):
            return self.propagate(edge_index, x=x, gate=gate)

        def message(self, x_j, x_i, gate):
            return gate * x_j - 0.5 * x_i

        def update(self, aggr_out):
            return aggr_out + 1.0

    x0, g0 = torch.randn(N, 12), torch.rand(E, 1)
    for aggr in ("add", "mean", "max"):
        cot = torch.randn(N, 12)

        def run(dt, aggr=aggr, cot=cot):
            xo = x0.to(dt).requires_grad_(True)
            o = O.scatter(g0.to(dt) * xo[b.edge_index[0]] - 0.5 * xo[b.edge_index[1]], b.edge_index[1], N, "sum" if aggr == "add" else aggr) + 1.0
            return o, _grads(o, cot.to(dt), [xo])
        x = x0.to(device).requires_grad_(True)
        out = EdgeGated(aggr)(x, ei, g0.to(device))
        assert_twin_parity(run, out, _grads(out, cot.to(device), [x]), f"user subclass aggr={aggr}", ["x"])
    with pytest.raises(TypeError):
        EdgeGated("add").propagate(ei, x=x0.to(device))              # message() needs `gate`


# ---------------------------------------------------------------------------------------------
# training mode of the reference's DEFAULT configuration (model.py:30-31: RReLU x 3, Dropout(0.2)): statistical parity
# ---------------------------------------------------------------------------------------------
def test_rrelu_and_dropout_device_stream_statistics(device):
    """torch.nn.RReLU(1/8, 1/3) in training mode multiplies every non-positive input by a ~ U(lower, upper); Dropout(p) zeroes with
    probability p and scales the rest by 1 / (1 - p).  The HIP ops draw from a device-side Philox stream: the numbers differ
    from torch's generator, so parity is statistical (moments, range, uniformity, independence between launches), the
    backward must use exactly the numbers of its forward, and a re-seeded stream must replay."""
    n = 1 << 20
    x = torch.full((n // 64, 64), -1.0, device=device, requires_grad=True)
    ops.manual_seed(1234)
    out = ops.rrelu(x)
    a = (-out).detach().flatten().double().cpu()                      # the slopes
    lo, hi = 1 / 8, 1 / 3
    assert a.min() >= lo and a.max() <= hi, (a.min().item(), a.max().item())
    assert abs(a.mean().item() - (lo + hi) / 2) < 3e-4, a.mean().item()   # sigma of the mean = 0.06 / 1024 = 6e-5
    assert abs(a.var().item() - (hi - lo) ** 2 / 12) < 3e-5, a.var().item()
    hist = torch.histc(a.float(), bins=16, min=lo, max=hi)
    chi2 = ((hist - n / 16) ** 2 / (n / 16)).sum().item()
    assert chi2 < 50, chi2                                            # 15 degrees of freedom: P(chi2 > 50) ~ 1e-5
    c1 = torch.corrcoef(torch.stack([a[:-1], a[1:]]))[0, 1].item()
    assert abs(c1) < 5e-3, c1                                         # neighbours are independent
    (g,) = torch.autograd.grad(out.sum(), [x])
    assert torch.equal(g.flatten().double().cpu(), a), "backward regenerates the slopes of its forward"
    out2 = ops.rrelu(x)                                               # next launch: a different stream position
    assert (out2 != out).float().mean().item() > 0.99, "the stream position did not advance"
    assert abs(torch.corrcoef(torch.stack([a, (-out2).detach().flatten().double().cpu()]))[0, 1].item()) < 5e-3
    ops.manual_seed(1234)
    assert torch.equal(ops.rrelu(x), out), "same seed, same stream position: same numbers"
    xp = torch.rand(1000, 60, device=device) + 0.1                    # positive inputs pass through untouched
    assert torch.equal(ops.rrelu(xp), xp)

    # inputs bounded away from zero: the test reads the mask off the output (a randn draw that is exactly 0 would look dropped)
    y = ((torch.rand(n // 64, 64, device=device) + 0.5) * torch.where(torch.rand(n // 64, 64, device=device) < 0.5, -1.0, 1.0)).requires_grad_(True)
    for p in (0.2, 0.5):
        d = ops.dropout(y, p)
        keep = (d != 0).double().mean().item()
        assert abs(keep - (1 - p)) < 2e-3, (p, keep)
        m = d.detach() != 0
        assert_close(d.detach()[m], (y.detach() / (1 - p))[m], 1e-6, "kept values are scaled by 1 / (1 - p)")
        (gd,) = torch.autograd.grad(d.sum(), [y])
        assert torch.equal(gd != 0, m) and abs(gd[m].double().mean().item() - 1 / (1 - p)) < 1e-6, "backward uses the forward's mask"
    # eval mode is deterministic and equals torch
    mod = torch.nn.RReLU().eval()
    blk = layer.LinearBlock(64, 64, act="RReLU").to(device).eval()
    assert_close(blk(y.detach()), mod(torch.nn.functional.linear(y.detach(), blk.linear.weight, blk.linear.bias)), 1e-5, "eval RReLU")


@pytest.mark.parametrize("alpha", [4, 2, 3])
def test_default_config_training_step_uses_the_fused_tails(device, monkeypatch, alpha):
    """Architecture() with NO overrides except the conv (model.py:24-33 defaults: RReLU x 3, graph_do = end_do = Dropout(0.2)) in
    train(): the block tail draws the RReLU slopes and writes the next step's dropped input itself, the RReLU launches behind
    mol_lin0 / mol_flat write the dropped inputs of message step 1 / lin_out1 (no standalone dropout launch, no torch RNG kernels); the analytic gradient matches a central difference of the
    re-seeded forward; eval() still equals the oracle.  Odd hidden widths (alpha 2, 3: 30 -> 32, 45 -> 48) take the same kernels on
    zero-padded rows and gate-wise padded GRU parameters."""
    torch.manual_seed(3)
    b = synth_batch(48, seed=8).to(device)
    net = model.Architecture(mol_block="_TripletMessage", e_dim=128, hid_dim_alpha=alpha).to(device)
    calls = {"dropout": 0, "rrelu": 0}
    real_drop, real_rrelu = ops.dropout, ops.rrelu
    monkeypatch.setattr(ops, "dropout", lambda x, p: (calls.__setitem__("dropout", calls["dropout"] + 1), real_drop(x, p))[1])
    monkeypatch.setattr(ops, "rrelu", lambda *a, **k: (calls.__setitem__("rrelu", calls["rrelu"] + 1), real_rrelu(*a, **k))[1])
    net.train()
    ops.manual_seed(99)
    out = net(b)
    # standalone launches: RReLU after mol_lin0 and mol_flat — each also writes the dropped twin for the Dropout that follows it (before
    # message step 1, before lin_out1), so no dropout launch of its own is left
    # (hid 60: mol_lin0's RReLU and its twin come out of the embedding product's own epilogue — glam_ts_gemm_rrelu / _act_node — so ONE
    # stand-alone RReLU launch is left)
    # (... and mol_flat's RReLU and the Dropout behind it are applied by the head as it reads: glam_linear_narrow_act_fwd — none left)
    fused_embedding = alpha == 4 and ops.RRELU_IN_GEMM and ops._lib.route_enabled("x3")
    assert calls == {"dropout": 0, "rrelu": 2 - int(fused_embedding) - int(ops.HEAD_ACT_FUSED)}, calls
    loss = out.square().mean()
    grads = torch.autograd.grad(loss, list(net.parameters()))
    ops.manual_seed(99)
    assert torch.equal(net(b), out), "re-seeded stream replays the training-mode forward"
    ops.manual_seed(98)
    assert not torch.equal(net(b), out)
    # directional derivative along a random parameter direction (same seed on both sides: the masks are those of `out`)
    torch.manual_seed(4)
    vs = [torch.randn_like(p) * p.abs().mean() for p in net.parameters()]

    def loss_at(step):
        with torch.no_grad():
            for p, v in zip(net.parameters(), vs):
                p.add_(v, alpha=step)
            ops.manual_seed(99)
            val = net(b).double().square().mean().item()
            for p, v in zip(net.parameters(), vs):
                p.sub_(v, alpha=step)
        return val
    an = sum((g.double() * v.double()).sum().item() for g, v in zip(grads, vs))
    # the loss is only piecewise smooth (the sort-pool's top-3 selection can flip inside a finite step): two of three step sizes
    # have to agree with the analytic value
    fds = [(loss_at(+eps) - loss_at(-eps)) / (2 * eps) for eps in (1e-2, 1e-3, 3e-4)]
    assert sum(abs(fd - an) <= 3e-2 * max(abs(an), abs(fd)) + 1e-6 for fd in fds) >= 2, (fds, an)
    # eval mode: deterministic, equals the oracle (RReLU -> its mean slope, dropout off)
    net.eval()
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    bc = b.to("cpu")
    ref = O.architecture(sd, bc, b.num_graphs, message_steps=3, mol_block="_TripletMessage", mol_readout="GlobalPool5")
    b64 = type(bc)(bc.x.double(), bc.edge_index, bc.edge_attr.double(), batch=bc.batch)
    ref64 = O.architecture({k: v.double() for k, v in sd.items()}, b64, b.num_graphs, message_steps=3, mol_block="_TripletMessage",
                           mol_readout="GlobalPool5")
    assert_fp32_parity(net(b), ref64, ref, "eval output", out_tol=1e-5)


def test_default_config_graphed_training_follows_the_eager_stream(device):
    """RReLU / Dropout numbers come from a stream position kept in device memory and advanced by the kernels themselves: replayed
    hipGraphs continue the stream exactly like eager steps, so the graphed and the eager run of the default (training-mode)
    configuration from one seed produce the same losses and parameters — and a batch sees fresh masks on every visit."""
    import copy
    from glam_amd.data import DataLoader, synth_molecule
    from glam_amd.graphs import GraphedTrainStep
    rng = np.random.default_rng(13)
    mols = [synth_molecule(rng) for _ in range(16)]
    torch.manual_seed(9)
    net0 = model.Architecture(mol_block="_TripletMessage", e_dim=64).to(device).train()
    loss_fn = lambda out, b: torch.nn.functional.mse_loss(out.view(-1), b.y.view(-1))
    results = []
    for graphed in (False, True):
        net = copy.deepcopy(net0)
        opt = torch.optim.Adam(net.parameters(), lr=2.0 ** -10, capturable=True)
        loader = DataLoader(mols, batch_size=8, device=device)
        stepper = GraphedTrainStep(net, opt, loss_fn)
        ops.manual_seed(77)
        losses = []
        for _epoch in range(4):
            for b in loader:
                if graphed:
                    losses.append(float(stepper(b)))
                else:
                    opt.zero_grad(set_to_none=True)
                    loss = loss_fn(net(b), b)
                    loss.backward()
                    opt.step()
                    losses.append(float(loss.detach()))
        results.append((losses, [p.detach().clone() for p in net.parameters()]))
    (l_e, p_e), (l_g, p_g) = results
    assert np.allclose(l_e, l_g, rtol=1e-5, atol=1e-6), (l_e, l_g)
    for a, r in zip(p_g, p_e):
        assert_close(a, r, 1e-5, "parameter")


# ---------------------------------------------------------------------------------------------
# software-pipelined forward aggregate (csrc/triplet_dma.hip): bit-identical to the general kernel
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("H,C,De,onehot", [(3, 60, 4, True), (3, 60, 4, False), (1, 60, 4, True), (2, 32, 4, False), (4, 44, 8, False),
                                           (3, 15, 3, True), (3, 64, 8, True)])
def test_pipelined_forward_is_bit_identical_to_the_general_kernel(device, H, C, De, onehot):
    """glam_triplet_fwd_ell (index records + rows prefetched one pass ahead, one-hot e_ij fast path) against glam_triplet_fwd on
    the same tensors: equal bit for bit — molecules with isolated atoms and every in-degree 0..4, N not a multiple of the pass
    size, both LDS-DMA and register variants of the pipeline.  Graphs with an in-degree above 4 report no ELL form."""
    from glam_amd import _lib
    lib, p, st = ops._lib.load(), ops._lib.ptr, ops._lib.stream
    torch.manual_seed(H * 100 + C)
    b = synth_batch(300, seed=C + H)
    N0 = b.x.size(0)
    ei = torch.cat([b.edge_index, torch.tensor([[5, 6, 7, 9], [8, 8, 8, 8]])], dim=1)      # node 8: in-degree 4 (2 + extra)... or more
    deg = torch.bincount(ei[1], minlength=N0)
    keep = torch.ones(ei.size(1), dtype=torch.bool)
    for n in (deg > 4).nonzero().flatten().tolist():                                       # trim any node back to in-degree 4
        idx = (ei[1] == n).nonzero().flatten()[4:]
        keep[idx] = False
    ei = ei[:, keep]
    N = N0 + 3                                                                              # three isolated atoms at the end: N % 4 != 0
    E = ei.size(1)
    Cp, Dp = (C + 3) // 4 * 4, 4 if De <= 4 else 8
    ea = torch.zeros(E, Dp)
    if onehot:
        ea[torch.arange(E), torch.randint(0, De, (E,))] = 1.0
    else:
        ea[:, :De] = torch.rand(E, De)
    ei, ea = ei.to(device), ea.to(device)
    xw, a_ij = torch.randn(N, H * Cp, device=device), torch.randn(N, 8, device=device)
    We, M = torch.randn(Dp, H * Cp, device=device), torch.randn(Dp, 4, device=device)
    gi = ops.GraphIndex(ei, N)
    ell = gi.ell()
    assert ell is not None and int((ell[0] >= 0).sum()) == E
    ref_a, ref_s = torch.empty(N, H * Cp, device=device), torch.empty(N, 8, device=device)
    _lib.check(lib.glam_triplet_fwd(p(xw), p(a_ij), p(ea), p(We), p(M), p(gi.rowptr), p(gi.src), p(gi.eid), N, E, H, Cp, Dp, 1, 0.2,
                                    p(ref_a), p(ref_s), st()), "fwd")
    assert lib.glam_triplet_fwd_ell_supported(H, Cp, Dp)
    for grid in (0, 3, 40):                                                                 # default grid; many passes per wave; one pass per wave
        got_a, got_s = torch.full_like(ref_a, 9.0), torch.full_like(ref_s, 9.0)
        _lib.check(lib.glam_triplet_fwd_ell(p(xw), p(a_ij), p(ea), p(We), p(M), p(ell[0]), p(ell[1]), N, E, H, Cp, Dp, 0.2, int(onehot),
                                            p(got_a), p(got_s), grid, st()), "fwd_ell")
        assert torch.equal(got_a, ref_a) and torch.equal(got_s, ref_s), f"grid={grid}"
    # an in-degree of 5: no ELL form, the op keeps the general kernel
    ei5 = torch.cat([ei, torch.tensor([[1, 2, 3], [8, 8, 8]], device=device)], dim=1)
    assert ops.GraphIndex(ei5, N).ell() is None


def test_pipelined_forward_serves_large_molecular_batches_through_the_op(device, monkeypatch):
    """ops.triplet_aggregate switches to the pipelined kernel above GraphIndex.ELL_MIN_NODES: same result, same gradients."""
    b = synth_batch(64, seed=2).to(device)
    N = b.x.size(0)
    torch.manual_seed(1)
    conv = layer.TripletMessage(60, 4).to(device)
    with torch.no_grad():
        Wn, Wa, We, M, Ws, Cp, Dp = conv._staged_weights()
    x = torch.randn(N, 60, device=device)
    outs = []
    for thr in (10 ** 9, 1):
        monkeypatch.setattr(ops.GraphIndex, "ELL_MIN_NODES", thr)
        xw = (x @ Wn).requires_grad_(True)
        a_ij = (x @ Wa).requires_grad_(True)
        gi = ops.GraphIndex(b.edge_index, N)
        aggr = ops.triplet_aggregate(xw, a_ij, b.edge_attr, We, M, gi, 3, Cp)
        g = torch.autograd.grad(aggr.square().sum(), [xw, a_ij])
        outs.append((aggr, g, gi._ell))
    assert outs[0][2] is False and outs[1][2] not in (False, None)                         # the ELL form was built only on the second route
    assert torch.equal(outs[0][0], outs[1][0]) and all(torch.equal(a, c) for a, c in zip(outs[0][1], outs[1][1]))


# ---------------------------------------------------------------------------------------------
# torch-extension front end (torch.ops.glam.*): same kernels behind at::Tensor operators with C++ autograd nodes
# ---------------------------------------------------------------------------------------------
def test_torch_extension_ops_match_the_ctypes_route(device, monkeypatch):
    from glam_amd import torch_ext
    G = torch_ext.load()
    torch.manual_seed(31)
    b = synth_batch(40, seed=31).to(device)
    N, E = b.x.size(0), b.edge_index.size(1)
    # CSR staging: both directions against GraphIndex
    gi = ops.GraphIndex(b.edge_index, N)
    rp, nb, ed, err = G.csr_from_edge_index(b.edge_index, N, 0)
    cp, ds, et, _ = G.csr_from_edge_index(b.edge_index, N, 1)
    colptr, dstv, eid_t = gi.transpose()
    for a_, r_ in ((rp, gi.rowptr), (nb, gi.src), (ed, gi.eid), (cp, colptr), (ds, dstv), (et, eid_t)):
        assert torch.equal(a_, r_)
    assert int(err) == 0 and int(G.csr_from_edge_index(b.edge_index, N - 5, 0)[3]) == 1
    ptr, perr = G.batch_ptr(b.batch, 40)
    assert torch.equal(ptr, ops.segment_ptr(b.batch, 40).ptr) and int(perr) == 0
    # whole layer: output and all six gradients, bit for bit (same launches)
    conv = layer.TripletMessage(60, 4).to(device)
    with torch.no_grad():
        conv.bias.normal_(0, 0.1)
    x0 = torch.randn(N, 60, device=device)
    cot = torch.randn(N, 60, device=device)
    res = []
    for ext in (False, True):
        monkeypatch.setattr(ops, "USE_TORCH_EXT", ext)
        x = x0.clone().requires_grad_(True)
        ea = b.edge_attr.detach().clone().requires_grad_(True)        # the bond features' gradient too (layer.py:36 takes any tensor)
        out = conv(x, b.edge_index, ea)
        res.append((out, torch.autograd.grad(out, [x, ea] + list(conv.parameters()), grad_outputs=cot)))
    assert type(res[1][0].grad_fn).__name__ != type(res[0][0].grad_fn).__name__          # a C++ node on the extension route
    assert float(res[1][1][1].abs().max()) > 0
    assert torch.equal(res[0][0], res[1][0])
    for a_, r_ in zip(res[1][1], res[0][1]):
        assert torch.equal(a_, r_)
    monkeypatch.setattr(ops, "USE_TORCH_EXT", False)
    # aggregate op (multi-head and light form)
    with torch.no_grad():
        Wn, Wa, We, M, Ws, Cp, Dp = conv._staged_weights()
    xw, a_ij = (x0 @ Wn).requires_grad_(True), (x0 @ Wa).requires_grad_(True)
    ref = ops.triplet_aggregate(xw, a_ij, b.edge_attr, We, M, gi, 3, Cp)
    got = G.triplet_aggregate(xw, a_ij, b.edge_attr, We, M, gi.rowptr, gi.src, gi.eid, colptr, dstv, eid_t, 3, 0.2)
    assert torch.equal(ref, got)
    c2 = torch.randn_like(ref)
    for a_, r_ in zip(torch.autograd.grad(got, [xw, a_ij], grad_outputs=c2), torch.autograd.grad(ref, [xw, a_ij], grad_outputs=c2)):
        assert torch.equal(a_, r_)
    xl = torch.randn(N, 60, device=device)
    refl = ops.light_aggregate(xl, a_ij.detach(), b.edge_attr, M, gi, 60)
    assert torch.equal(refl, G.triplet_aggregate(xl, a_ij.detach(), b.edge_attr, None, M, gi.rowptr, gi.src, gi.eid, colptr, dstv, eid_t, 1, 0.2))
    # readouts
    sp = ops.segment_ptr(b.batch, 40)
    h = torch.randn(N, 60, device=device, requires_grad=True)
    for mode, name in ((0, "sum"), (1, "mean"), (2, "max")):
        r_, g_ = ops.segment_pool(h, sp, name), G.segment_pool(h, sp.ptr, mode)
        assert torch.equal(r_, g_)
        assert torch.equal(torch.autograd.grad(r_.sum(), [h])[0], torch.autograd.grad(g_.sum(), [h])[0])
    gate = torch.randn(N, device=device, requires_grad=True)
    assert torch.equal(ops.segment_attention(gate, h, sp), G.segment_softmax_aggregate(gate, h, sp.ptr))
    p5 = ops.pool5(h, sp, 3)
    assert torch.equal(p5, G.global_pool5(h, sp.ptr, 3)) and torch.equal(p5[:, 120:], G.sort_pool_topk_last(h, sp.ptr, 3))
    assert torch.equal(torch.autograd.grad(p5.square().sum(), [h])[0], torch.autograd.grad(G.global_pool5(h, sp.ptr, 3).square().sum(), [h])[0])
    # misuse: RuntimeError (TORCH_CHECK), the device stays usable
    with pytest.raises(RuntimeError):
        G.segment_pool(h.double(), sp.ptr, 0)
    with pytest.raises(RuntimeError):
        G.triplet_layer(x0[:, :30].contiguous(), b.edge_attr, *conv.parameters(), gi.rowptr, gi.src, gi.eid, colptr, dstv, eid_t, 3, 0.2)
    assert torch.isfinite(G.segment_pool(h, sp.ptr, 0)).all()


@pytest.mark.parametrize("C,B", [(60, 1024), (60, 5), (45, 150), (36, 40), (64, 33)])
def test_b1_matrix_waves_on_presplit_fragments(device, monkeypatch, C, B):
    """The warp-specialised backward by target reads W_scale^T as operand fragments split into their three bf16 terms by the staging
    launch (Staged::dagg_pre: one launch per weight update) instead of splitting the plain image in every block's prologue
    (``GLAM_B1_PRE=0``): the same split function on the same numbers — every gradient bit for bit."""
    b = synth_batch(B, seed=B + C).to(device)
    torch.manual_seed(C)
    conv = layer.TripletMessage(C, 4, heads=3).to(device)
    N = b.x.size(0)
    x0, cot = torch.randn(N, C, device=device), torch.randn(N, C, device=device)
    monkeypatch.setattr(ops, "USE_TORCH_EXT", False)
    res = []
    for pre in ("1", "0"):
        monkeypatch.setenv("GLAM_B1_PRE", pre)
        x = x0.clone().requires_grad_(True)
        out = conv(x, b.edge_index, b.edge_attr)
        res.append(torch.autograd.grad(out, [x] + list(conv.parameters()), grad_outputs=cot))
    for u, v in zip(*res):
        assert torch.equal(u, v)


@pytest.mark.parametrize("C,H,B", [(60, 3, 1024), (60, 3, 7), (40, 4, 200), (64, 2, 90), (45, 3, 150), (60, 1, 64), (36, 3, 100), (52, 2, 33), (37, 1, 20)])
def test_warp_specialised_forward_equals_the_general_fused_kernel(device, monkeypatch, C, H, B):
    """csrc/triplet_ws.hip / triplet_ws_b1.hip (producer waves gather, consumer waves run the fused GEMM out of an LDS tile ring; what the
    op launches for molecular graphs at every size) against the general fused kernels: output and the gradients of x, weight_node,
    weight_scale and bias equal bit for bit — every head count, padded widths, tile counts that are odd / smaller than the ring / not a
    multiple of the producer groups, isolated atoms; the warp-specialised B1 (H <= 3) sums the d_W_edge / d_M block partials in another
    fixed order (rounding-level differences in weight_edge and the edge third of weight_triplet_att).  ``ops.WS_ROUTE = "0"`` and
    ``GLAM_WS=0`` both select the general kernels."""
    b = synth_batch(B, seed=B + C).to(device)
    torch.manual_seed(C + H)
    conv = layer.TripletMessage(C, 4, heads=H).to(device)
    with torch.no_grad():
        conv.bias.normal_(0, 0.1)
    N = b.x.size(0)
    x0 = torch.randn(N, C, device=device)
    cot = torch.randn(N, C, device=device)
    from glam_amd import _lib
    lib = _lib.load()
    Cp = (C + 3) // 4 * 4
    assert lib.glam_triplet_layer_ws_supported(H, Cp, 4, 1) == 1
    assert lib.glam_triplet_layer_ws_supported(H, Cp, 4, 0) == 0 and lib.glam_triplet_layer_ws_supported(H, Cp, 8, 1) == 0
    monkeypatch.setenv("GLAM_TORCH_EXT", "0")
    monkeypatch.setattr(ops, "USE_TORCH_EXT", False)
    res = {}
    # x3 = "0": every dense product on the fp32 matrix instructions (the data flow of the warp-specialised kernels is pinned bit for bit
    # against the general ones); x3 = "1" (the default): the consumers' products in 3 x bf16 form (csrc/bf16x3.h) — fp32 accuracy, other
    # roundings: rounding-level agreement here, the fp64-twin bound against the oracle in the tests around this one
    for name, route, env, x3 in (("general", "0", "1", "0"), ("general_env", "auto", "0", "0"), ("ws", "auto", "1", "0"), ("ws_x3", "auto", "1", "1")):
        monkeypatch.setattr(ops, "WS_ROUTE", route)
        monkeypatch.setenv("GLAM_WS", env)
        monkeypatch.setenv("GLAM_X3", x3)
        x = x0.clone().requires_grad_(True)
        with _lib.kernel_timer(capacity=64) as kt:
            out = conv(x, b.edge_index, b.edge_attr)
            grads = torch.autograd.grad(out, [x] + list(conv.parameters()), grad_outputs=cot)
        res[name] = (out, grads, [n for n, _, _ in kt.records()])
    for nm in ("ws", "ws_x3"):
        for k in ("k_triplet_fwd_ws", "k_triplet_bwd_src_ws") + (("k_triplet_bwd_dst_ws",) if H <= 3 else ()):
            assert any(k in n for n in res[nm][2]), (k, res[nm][2])
    assert not any("_ws" in n for n in res["general"][2] + res["general_env"][2])
    assert torch.equal(res["general"][0], res["general_env"][0]) and all(torch.equal(a, c) for a, c in zip(res["general"][1], res["general_env"][1]))
    names = ["x", "weight_node", "weight_edge", "weight_triplet_att", "weight_scale", "bias"]
    assert torch.equal(res["general"][0], res["ws"][0]), (res["general"][0] - res["ws"][0]).abs().max().item()
    for pn, a, c in zip(names, res["general"][1], res["ws"][1]):
        if H <= 3 and pn in ("weight_edge", "weight_triplet_att"):
            # d_x being bit-equal pins every per-edge and per-node quantity of B1 (alpha_e, dpre_e, d_a_i, d_aggr)
            assert_close(c, a, 2e-6, f"ws d_{pn}")
        else:
            assert torch.equal(a, c), (pn, (a - c).abs().max().item())
    assert_close(res["ws_x3"][0], res["general"][0], 2e-6, "ws_x3 out")
    for pn, a, c in zip(names, res["general"][1], res["ws_x3"][1]):
        assert_close(c, a, 4e-6, f"ws_x3 d_{pn}")


# ---------------------------------------------------------------------------------------------
# narrow-output linear (the model's output head, model.py:47,61): row dot products instead of a library GEMM
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,K,M,bias", [(1024, 1024, 1, True), (1024, 1024, 2, True), (37, 300, 12, True), (1, 64, 16, False),
                                        (2039, 1024, 1, True), (640, 1024, 5, False), (200, 1024, 2, True), (5000, 300, 3, True)])
def test_narrow_linear_against_the_fp64_twin(device, N, K, M, bias):
    from tests.conftest import assert_fp32_parity
    g = torch.Generator().manual_seed(N * 7 + K + M)
    x = torch.randn(N, K, generator=g)
    w = torch.randn(M, K, generator=g) / K ** 0.5
    b = torch.randn(M, generator=g) if bias else None
    cot = torch.randn(N, M, generator=g)

    def run(dt, dev, fn):
        xs = x.detach().clone().to(dev, dt).requires_grad_(True)
        ws = w.detach().clone().to(dev, dt).requires_grad_(True)
        bs = None if b is None else b.detach().clone().to(dev, dt).requires_grad_(True)
        y = fn(xs, ws, bs)
        y.backward(cot.to(dev, dt))
        return [y, xs.grad, ws.grad] + ([] if bs is None else [bs.grad])

    ref64 = run(torch.float64, "cpu", torch.nn.functional.linear)
    ref32 = run(torch.float32, "cpu", torch.nn.functional.linear)
    assert ops._lib.load().glam_linear_narrow_supported(K, M)
    got = run(torch.float32, device, ops.linear)
    for name, a, r64, r32 in zip(["y", "d_x", "d_w", "d_b"], got, ref64, ref32):
        assert_fp32_parity(a, r64, r32, f"narrow linear {N}x{K}->{M} {name}", out_tol=1e-5 if name == "y" else None)
    # bit-reproducible (fixed-order sums), and an input without gradient skips d_x
    again = run(torch.float32, device, ops.linear)
    assert all(torch.equal(a, c) for a, c in zip(got, again))
    xs = x.detach().clone().to(device)
    ws = w.detach().clone().to(device).requires_grad_(True)
    ops.linear(xs, ws, None if b is None else b.to(device)).backward(cot.to(device))
    assert torch.equal(ws.grad, got[2])


def test_narrow_linear_is_what_the_output_head_runs(device, monkeypatch):
    calls = []
    real = ops._LinearNarrow.apply
    monkeypatch.setattr(ops._LinearNarrow, "apply", staticmethod(lambda *a: (calls.append(a[1].shape), real(*a))[1]))
    b = synth_batch(8, seed=1).to(device)
    net = model.Architecture(mol_block="_TripletMessage", out_dim=2).to(device).eval()
    out = net(b)
    assert out.shape == (8, 2) and calls == [torch.Size([2, 1024])]
    raw = ops._lib.load()
    assert not raw.glam_linear_narrow_supported(1024, 17) and not raw.glam_linear_narrow_supported(1022, 1)


@pytest.mark.parametrize("N,D", [(1024, 1024), (5000, 300), (1, 64), (40000, 64), (0, 8)])
def test_colsum_and_the_library_linear_node(device, N, D):
    from tests.conftest import assert_fp32_parity
    g = torch.Generator().manual_seed(N + D)
    x = torch.randn(N, D, generator=g)
    raw = ops._lib.load()
    xd, out = x.to(device), torch.full((D,), float("nan"), device=device)
    ws = torch.empty(max(raw.glam_colsum_workspace_bytes(D), 16), dtype=torch.uint8, device=device)
    ops.check(raw.glam_colsum(ops.ptr(xd), N, D, D, ops.ptr(out), ops.ptr(ws), ws.numel(), ops.stream()), "glam_colsum")
    assert_fp32_parity(out, x.double().sum(0), x.sum(0), f"colsum {N}x{D}")
    out2 = torch.empty_like(out)
    ops.check(raw.glam_colsum(ops.ptr(xd), N, D, D, ops.ptr(out2), ops.ptr(ws), ws.numel(), ops.stream()), "glam_colsum")
    assert torch.equal(out, out2)
    if N == 0:
        return
    # the readout MLP's linear (K = 300 > 192: matrix products on the library): same numbers as F.linear, bias gradient from colsum
    K = 300
    a = torch.randn(min(N, 2048), K, generator=g)
    w, b = torch.randn(D, K, generator=g) / K ** 0.5, torch.randn(D, generator=g)
    cot = torch.randn(a.size(0), D, generator=g)

    def run(dt, dev, fn):
        t = [v.detach().clone().to(dev, dt).requires_grad_(True) for v in (a, w, b)]
        fn(*t).backward(cot.to(dev, dt))
        return [v.grad for v in t]

    ref64, ref32 = run(torch.float64, "cpu", torch.nn.functional.linear), run(torch.float32, "cpu", torch.nn.functional.linear)
    got = run(torch.float32, device, ops.linear)
    for name, v, r64, r32 in zip(["d_x", "d_w", "d_b"], got, ref64, ref32):
        assert_fp32_parity(v, r64, r32, f"library linear {a.size(0)}x{K}->{D} {name}")


@pytest.mark.parametrize("N", [105, 20400])
@pytest.mark.parametrize("K,M", [(60, 180), (180, 60), (276, 92)])
def test_ts_gemm_pair_equals_two_launches(device, N, K, M):
    """Two products in one launch (the GRU's gate linears / their input gradients) are bit-identical to the two single launches, with every
    per-product option: CELU on the operand, bias, CELU' of a source on the output, an addend."""
    raw, p, st = ops._lib.load(), ops.ptr, ops.stream
    g = torch.Generator().manual_seed(N + K)
    r = lambda *s: torch.randn(*s, generator=g).to(device)
    x, h, wa, wb, ba, bb, src, add = r(N, K), r(N, K), r(M, K), r(M, K), r(M), r(M), r(N, M), r(N, M)
    nb = raw.glam_ts_gemm_image_bytes(K, M) // 4
    ia, ib = torch.empty(nb, device=device), torch.empty(nb, device=device)
    ops.check(raw.glam_ts_gemm_make_image(p(wa), K, 1, K, M, p(ia), st()), "image")
    ops.check(raw.glam_ts_gemm_make_image(p(wb), K, 1, K, M, p(ib), st()), "image")
    o1, o2, q1, q2 = (torch.empty(N, M, device=device) for _ in range(4))
    ops.check(raw.glam_ts_gemm_celu(p(x), K, K, 1, p(ia), p(ba), p(o1), M, M, p(src), M, N, st()), "celu")
    ops.check(raw.glam_ts_gemm_add(p(h), K, K, p(ib), p(bb), p(o2), M, M, p(add), M, N, st()), "add")
    ops.check(raw.glam_ts_gemm_pair(p(x), K, K, 1, p(ia), p(ba), p(q1), M, M, p(src), M, None, 0,
                                    p(h), K, K, 0, p(ib), p(bb), p(q2), M, M, None, 0, p(add), M, N, st()), "pair")
    assert torch.equal(o1, q1) and torch.equal(o2, q2)
    ref = torch.nn.functional.linear(h.double().cpu(), wb.double().cpu(), bb.double().cpu()) + add.double().cpu()
    assert (q2.double().cpu() - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    # the two products must share N and the kernel variant
    img_other = torch.empty(raw.glam_ts_gemm_image_bytes(M, K) // 4, device=device)
    assert raw.glam_ts_gemm_pair(p(x), K, K, 0, p(ia), None, p(q1), M, M, None, 0, None, 0,
                                 p(o1), M, M, 0, p(img_other), None, p(torch.empty(N, K, device=device)), K, K, None, 0, None, 0, N, st()) != 0


@pytest.mark.parametrize("C,train", [(60, False), (60, True), (32, False), (48, True), (24, True)])
def test_warp_specialised_gru_forward_against_the_fp32_launches(device, monkeypatch, C, train):
    """glam_gru_ws_fwd / _rng_fwd (the default GRU step: gate products in 3 x bf16 form, gates + residual + activation (+ RReLU / Dropout)
    in the consumers' epilogue) against glam_ts_gemm_pair + glam_gru_tail_* on the fp32 matrix cores: the same Philox words (identical
    Dropout masks), outputs and every gradient equal to rounding (the gradients run through the same backward kernels on gi / gh that
    differ in their last bits)."""
    b = synth_batch(40, seed=C).to(device)
    blk = layer.MessageBlock(C, C, 4, norm="_None", dropout="Dropout(0.2)" if train else "_None()", conv="_TripletMessage",
                             act="RReLU" if train else "CELU", res=True).to(device)
    blk.train(train)
    x0 = torch.randn(b.x.size(0), C, device=device)
    assert ops._lib.load().glam_gru_ws_supported(C) == 1 and not ops._lib.load().glam_gru_ws_supported(20) and not ops._lib.load().glam_gru_ws_supported(68)
    res = []
    for ws in ("0", "1"):
        monkeypatch.setattr(ops, "GRU_WS", ws)
        monkeypatch.setattr(ops, "GRU_FUSED", "0")
        ops.manual_seed(11, device)
        x = x0.clone().requires_grad_(True)
        with ops.weight_scope():
            x1, h1 = blk(x, b.edge_index, b.edge_attr, h=None, batch=b.batch)
            xd = layer._apply_dropout(blk.dropout, x1)            # the dropped twin the tail wrote (training mode)
            x2, h2 = blk(x1, b.edge_index, b.edge_attr, h=h1, batch=b.batch)
            gs = torch.autograd.grad((x2 * x2).sum() + h2.sum(), [x] + list(blk.parameters()))
        res.append([x1, xd, x2, h2] + list(gs))
    if train:
        assert torch.equal(res[0][1] == 0, res[1][1] == 0), "same Philox words: the same Dropout mask"
    for i, (a, c) in enumerate(zip(*res)):
        assert_close(c, a, 1e-5 if i >= 4 else 2e-6, f"ws vs fp32 launches, tensor {i}")


@pytest.mark.parametrize("N,C,ident,celu", [(1, 64, True, True), (1000, 64, False, False), (20400, 60, True, True), (17, 24, True, False), (0, 60, True, True)])
def test_gru_ws_fwd_c_abi(device, N, C, ident, celu):
    """glam_gru_ws_fwd called directly (widths up to 64, ragged and empty row counts, with / without residual and folded CELU) against the
    fp64 gate equations of torch.nn.GRU (src_1gp/layer.py:261-266)."""
    raw, p, st = ops._lib.load(), ops.ptr, ops.stream
    g = torch.Generator().manual_seed(N + C)
    r = lambda *s: torch.randn(*s, generator=g).to(device)
    x, h, idn, w_ih, w_hh, b_ih, b_hh = r(N, C), r(N, C), r(N, C), r(3 * C, C) * 0.3, r(3 * C, C) * 0.3, r(3 * C), r(3 * C)
    nb = raw.glam_ts_gemm_image_bytes(C, 3 * C) // 4
    ia, ib = torch.empty(nb, device=device), torch.empty(nb, device=device)
    ops.check(raw.glam_ts_gemm_make_image(p(w_ih), C, 1, C, 3 * C, p(ia), st()), "image")
    ops.check(raw.glam_ts_gemm_make_image(p(w_hh), C, 1, C, 3 * C, p(ib), st()), "image")
    gi, gh = torch.full((N, 3 * C), float("nan"), device=device), torch.full((N, 3 * C), float("nan"), device=device)
    hn, out = torch.full((N, C), float("nan"), device=device), torch.full((N, C), float("nan"), device=device)
    ops.check(raw.glam_gru_ws_fwd(p(x), p(h), p(idn) if ident else None, p(ia), p(ib), p(b_ih), p(b_hh), N, C, int(celu), 1, 0.0,
                                  p(gi), p(gh), p(hn), p(out), st()), "glam_gru_ws_fwd")
    xd = torch.nn.functional.celu(x.double().cpu()) if celu else x.double().cpu()
    gi_r = xd @ w_ih.double().cpu().t() + b_ih.double().cpu()
    gh_r = h.double().cpu() @ w_hh.double().cpu().t() + b_hh.double().cpu()
    rr, zz = torch.sigmoid(gi_r[:, :C] + gh_r[:, :C]), torch.sigmoid(gi_r[:, C:2 * C] + gh_r[:, C:2 * C])
    nn_ = torch.tanh(gi_r[:, 2 * C:] + rr * gh_r[:, 2 * C:])
    hn_r = (1 - zz) * nn_ + zz * h.double().cpu()
    out_r = torch.relu(hn_r + (idn.double().cpu() if ident else 0))
    for got, ref, what in ((gi, gi_r, "gi"), (gh, gh_r, "gh"), (hn, hn_r, "h_new"), (out, out_r, "out")):
        assert_close(got, ref, 2e-6, f"gru_ws {what} N={N} C={C}")
    assert raw.glam_gru_ws_fwd(p(x), p(h), None, p(ia), p(ib), p(b_ih), p(b_hh), N, 20, 0, 1, 0.0, p(gi), p(gh), p(hn), p(out), st()) != 0


@pytest.mark.parametrize("C", [60, 45, 32])
def test_skip_connection_through_the_conv_node(device, monkeypatch, C):
    """MessageBlock hands its skip connection (layer.py:253, 264) through the TripletMessage autograd node: the d_x product's epilogue adds
    d_identity (glam_triplet_layer_bwd_params_ell_add; C = 60, 45: warp-specialised route, 45 on zero-padded rows) — or the node adds it
    behind the general kernels (C = 32) — instead of autograd.  Same forward, the same gradient terms in another order of summation; the
    first application's GRU state shares the handed-back tensor."""
    b = synth_batch(40, seed=C).to(device)
    blk = layer.MessageBlock(C, C, 4, norm="_None", dropout="_None()", conv="_TripletMessage", act="ReLU", res=True).to(device)
    x0 = torch.randn(b.x.size(0), C, device=device)
    res = []
    for on in (False, True):
        monkeypatch.setattr(ops, "SKIP_THROUGH_CONV", on)
        x = x0.clone().requires_grad_(True)
        with ops.weight_scope():
            x1, h1 = blk(x, b.edge_index, b.edge_attr, h=None, batch=b.batch)
            x2, h2 = blk(x1, b.edge_index, b.edge_attr, h=h1, batch=b.batch)
            gs = torch.autograd.grad((x2 * x2).sum() + h2.sum(), [x] + list(blk.parameters()))
        res.append([x1, x2, h2] + list(gs))
    for i, (a, c) in enumerate(zip(*res)):
        if i < 3:
            assert torch.equal(a, c)                      # the forward is the same launches
        else:
            assert_close(c, a, 2e-6, f"gradient {i - 3}")  # the same terms, summed in another order (x of the first application gets three)
    # the C entry point refuses an addend where no warp-specialised backward by source runs (covered above through the op; here the error)
    raw = ops._lib.load()
    assert raw.glam_triplet_layer_ws_supported(3, 32, 4, 1) == 0 and raw.glam_triplet_layer_ws_supported(3, 48, 4, 1) == 1


@pytest.mark.parametrize("ragged", [False, True])
def test_layer_backward_adds_the_skip_gradient_in_the_d_x_epilogue(device, ragged):
    """_TripletLayer(with_identity=True): the gradient arriving at the handed-back input is added to d_x inside k_triplet_bwd_src_ws<ADD>
    (full tiles with the counted wait, the ragged last tile behind the loop) — bit for bit the separate add."""
    b = next(bb for bb in (synth_batch(24, seed=sd) for sd in range(200)) if (bb.x.size(0) % 16 != 0) == ragged).to(device)
    N = b.x.size(0)
    torch.manual_seed(N)
    conv = layer.TripletMessage(60, 4).to(device)
    x = torch.randn(N, 60, device=device, requires_grad=True)
    cot, cot2 = torch.randn(N, 60, device=device), torch.randn(N, 60, device=device)
    gi = ops.graph_index(b.edge_index, N)
    args = (b.edge_attr, conv.weight_node, conv.weight_edge, conv.weight_triplet_att, conv.weight_scale, conv.bias, gi, 3, 0.2)
    with ops.weight_scope():      # (the Python node: the route a captured step takes)
        out, ident = ops._TripletLayer.apply(x, *args, None, True)
        assert ident.data_ptr() == x.data_ptr() and torch.equal(out, ops._TripletLayer.apply(x, *args))
        (g_both,) = torch.autograd.grad((out * cot).sum() + (ident * cot2).sum(), [x], retain_graph=True)
        (g_conv,) = torch.autograd.grad((out * cot).sum(), [x], retain_graph=True)
        (g_skip,) = torch.autograd.grad((ident * cot2).sum(), [x])
    assert torch.equal(g_skip, cot2)
    assert torch.equal(g_both, g_conv + cot2)


@pytest.mark.parametrize("N,C,celu,hstate,ident", [(1, 64, True, True, True), (1000, 64, False, False, False), (20400, 60, True, True, True),
                                                   (17, 24, False, True, True), (0, 60, True, True, True)])
def test_gru_bwd_ws_c_abi(device, N, C, celu, hstate, ident):
    """glam_gru_bwd_ws (gate gradients + both input-gradient products in one launch) against glam_gru_tail_bwd + glam_ts_gemm_pair on the
    fp32 matrix cores: d_gi, d_gh, d_identity bit for bit (the same gate arithmetic), d_x and the complete d_h to rounding."""
    raw, p, st = ops._lib.load(), ops.ptr, ops.stream
    g = torch.Generator().manual_seed(N + C + 1)
    r = lambda *s: torch.randn(*s, generator=g).to(device)
    M = 3 * C
    gi, gh, h, out, d_out, d_hs, x, w_ih, w_hh = r(N, M), r(N, M), r(N, C), r(N, C), r(N, C), r(N, C), r(N, C), r(M, C) * 0.3, r(M, C) * 0.3
    nb = raw.glam_ts_gemm_image_bytes(M, C) // 4
    ta, tb = torch.empty(nb, device=device), torch.empty(nb, device=device)
    ops.check(raw.glam_ts_gemm_make_image(p(w_ih), C, 0, M, C, p(ta), st()), "image")
    ops.check(raw.glam_ts_gemm_make_image(p(w_hh), C, 0, M, C, p(tb), st()), "image")
    f = lambda *s: torch.full(s, float("nan"), device=device)
    # reference: the two-launch sequence with the products on the fp32 matrix cores
    dgi0, dgh0, dh0, did0, dx0, dhf0 = f(N, M), f(N, M), f(N, C), f(N, C), f(N, C), f(N, C)
    os.environ["GLAM_TALL_X3"] = "0"
    try:
        ops.check(raw.glam_gru_tail_bwd(p(gi), p(gh), p(h), p(out), p(d_out), p(d_hs) if hstate else None, N, C, 1, 0.0, p(dgi0), p(dgh0), p(dh0),
                                        p(did0) if ident else None, st()), "tail_bwd")
        ops.check(raw.glam_ts_gemm_pair(p(dgi0), M, M, 0, p(ta), None, p(dx0), C, C, p(x) if celu else None, C, None, 0,
                                        p(dgh0), M, M, 0, p(tb), None, p(dhf0), C, C, None, 0, p(dh0), C, N, st()), "pair")
    finally:
        del os.environ["GLAM_TALL_X3"]
    dgi, dgh, did, dx, dh = f(N, M), f(N, M), f(N, C), f(N, C), f(N, C)
    ops.check(raw.glam_gru_bwd_ws(p(gi), p(gh), p(h), p(out), p(d_out), p(d_hs) if hstate else None, p(x), p(ta), p(tb), N, C, int(celu), 1, 0.0, 0,
                                  p(dgi), p(dgh), p(did) if ident else None, p(dx), p(dh), st()), "glam_gru_bwd_ws")
    assert torch.equal(dgi, dgi0) and torch.equal(dgh, dgh0) and (not ident or torch.equal(did, did0))
    assert_close(dx, dx0, 2e-6, "d_x")
    assert_close(dh, dhf0, 2e-6, "d_h")
    if ident:       # merge_identity: d_h additionally carries d_identity (skip connection and GRU state are one tensor), d_identity untouched
        dh2, did2 = f(N, C), f(N, C)
        ops.check(raw.glam_gru_bwd_ws(p(gi), p(gh), p(h), p(out), p(d_out), p(d_hs) if hstate else None, p(x), p(ta), p(tb), N, C, int(celu), 1, 0.0, 1,
                                      p(dgi), p(dgh), p(did2), p(dx), p(dh2), st()), "glam_gru_bwd_ws merge")
        assert_close(dh2, dhf0 + did0, 2e-6, "d_h + d_identity")
        assert N == 0 or torch.isnan(did2).all()
    assert raw.glam_gru_bwd_ws(p(gi), p(gh), p(h), p(out), p(d_out), None, p(x), p(ta), p(tb), N, 20, 0, 1, 0.0, 0, p(dgi), p(dgh), None, p(dx), p(dh), st()) != 0
    if N:
        assert raw.glam_gru_bwd_ws(p(gi), p(gh), p(h), p(out), None, None, p(x), p(ta), p(tb), N, C, 0, 1, 0.0, 0, p(dgi), p(dgh), None, p(dx), p(dh), st()) != 0


@pytest.mark.parametrize("C,train", [(60, False), (60, True), (32, False), (48, True)])
def test_fused_gru_forward_is_bit_identical_to_the_two_launch_sequence(device, monkeypatch, C, train):
    """Opt-in glam_gru_fused_fwd (gate GEMMs + gates + residual + activation (+ RReLU / Dropout) in one launch) against
    glam_ts_gemm_pair + glam_gru_tail_*: same outputs, same gradients, same RNG stream."""
    monkeypatch.setattr(ops, "GRU_WS", "0")
    b = synth_batch(24, seed=C).to(device)
    blk = layer.MessageBlock(C, C, 4, norm="_None", dropout="Dropout(0.2)" if train else "_None()", conv="_TripletMessage",
                             act="RReLU" if train else "CELU", res=True).to(device)
    blk.train(train)
    x0 = torch.randn(b.x.size(0), C, device=device)
    res = []
    for fused in ("0", "1"):
        monkeypatch.setattr(ops, "GRU_FUSED", fused)
        ops.manual_seed(11, device)
        x = x0.clone().requires_grad_(True)
        with ops.weight_scope():
            x1, h1 = blk(x, b.edge_index, b.edge_attr, h=None, batch=b.batch)
            x2, h2 = blk(x1, b.edge_index, b.edge_attr, h=h1, batch=b.batch)
            gs = torch.autograd.grad((x2 * x2).sum() + h2.sum(), [x] + list(blk.parameters()))
        res.append([x1, x2, h2] + list(gs))
    for a, c in zip(*res):
        assert torch.equal(a, c)
    assert not ops._lib.load().glam_gru_fused_supported(66) and not ops._lib.load().glam_gru_fused_supported(0)


def test_graphed_epochs_with_several_steps_per_graph_launch(device):
    """GraphedTrainStep.run: consecutive steps of the cached loader captured into one graph per chunk (first epoch eager, second
    captured, later replayed) — the per-batch stepper's trajectory bit for bit, training-mode RReLU / Dropout stream included, and an
    lr change between epochs followed."""
    import copy
    from glam_amd.data import DataLoader, synth_molecule
    from glam_amd.graphs import GraphedTrainStep
    rng = np.random.default_rng(12)
    mols = [synth_molecule(rng) for _ in range(40)]
    torch.manual_seed(6)
    net0 = model.Architecture(mol_block="_TripletMessage", message_steps=2, mol_readout="GlobalPool5", e_dim=64).to(device).train()
    loss_fn = lambda out, b: torch.nn.functional.mse_loss(out.view(-1), b.y.view(-1))
    results = []
    for multi in (False, True):
        net = copy.deepcopy(net0)
        opt = torch.optim.Adam(net.parameters(), lr=2.0 ** -10, capturable=True)
        loader = DataLoader(mols, batch_size=8, device=device)          # 5 cached batches
        stepper = GraphedTrainStep(net, opt, loss_fn)
        ops.manual_seed(21, device)
        losses = []
        for epoch in range(4):
            if epoch == 3:
                opt.param_groups[0]["lr"] = 2.0 ** -12                   # what ReduceLROnPlateau does between epochs
            if multi:
                losses.append(stepper.run(loader, steps_per_graph=3))   # chunks of 3 + 2 steps
            else:
                losses.append(torch.stack([stepper(b).reshape(()) for b in loader]))
        if multi:
            assert stepper.graphs() == 2, stepper.graphs()
        results.append((torch.cat(losses), [p.detach().clone() for p in net.parameters()]))
    (l_s, p_s), (l_m, p_m) = results
    assert torch.equal(l_s, l_m)
    assert all(torch.equal(a, b) for a, b in zip(p_s, p_m))


def test_adam_one_launch_matches_the_library_optimizer(device):
    """``glam_amd.optim.Adam`` (one HIP launch over all parameter tensors; the optimizer of trainer.py:49-50) against
    ``torch.optim.Adam`` on the same gradient sequence: 25 steps, ragged tensor sizes (1 element, not a multiple of 4, several chunks),
    weight decay, a learning rate a scheduler changes on the way, a parameter that gets no gradient in some steps.  Tolerance: 2e-6
    relative to the largest parameter (fp32 re-association of the same formula; the moments to 1e-6)."""
    from glam_amd import optim
    torch.manual_seed(3)
    shapes = [(1,), (7,), (60, 180), (1024, 300), (3, 5, 11), (1023,), (4097,)]
    for wd in (0.0, 1e-2):
        ref = [torch.nn.Parameter(torch.randn(*s, device=device)) for s in shapes]
        mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
        o_ref = torch.optim.Adam(ref, lr=1e-2, weight_decay=wd, foreach=False)
        o_mine = optim.Adam(mine, lr=1e-2, weight_decay=wd)
        for step in range(25):
            if step == 10:
                for o in (o_ref, o_mine):
                    o.param_groups[0]["lr"] = 3e-3               # what ReduceLROnPlateau does (trainer.py:85)
            for p, q in zip(ref, mine):
                g = torch.randn_like(p) * (10.0 ** float(torch.randint(-3, 2, (1,))))
                p.grad, q.grad = g, g.clone()
            o_ref.step(); o_mine.step()
        for p, q in zip(ref, mine):
            assert_close(q, p, 2e-6, f"adam wd={wd} param {tuple(p.shape)}")
            assert_close(o_mine.state[q]["exp_avg"], o_ref.state[p]["exp_avg"], 1e-6, "exp_avg")
            assert_close(o_mine.state[q]["exp_avg_sq"], o_ref.state[p]["exp_avg_sq"], 1e-6, "exp_avg_sq")
            assert float(o_mine.state[q]["step"]) == 25.0 == float(o_ref.state[p]["step"])
    # state_dict round trip into a fresh optimizer: the trajectory continues as if nothing happened
    sd = o_mine.state_dict()
    clone = [torch.nn.Parameter(q.detach().clone()) for q in mine]
    o_clone = optim.Adam(clone, lr=1.0)
    o_clone.load_state_dict(sd)
    assert o_clone.param_groups[0]["lr"] == 3e-3
    for q, c in zip(mine, clone):
        g = torch.randn_like(q)
        q.grad, c.grad = g, g.clone()
    o_mine.step(); o_clone.step()
    for q, c in zip(mine, clone):
        assert torch.equal(q, c)
    assert float(o_clone.state[clone[0]]["step"]) == 26.0
    # a parameter without a gradient sits the step out (its values and moments untouched), the others move
    before = [q.detach().clone() for q in mine]
    mine[2].grad = None
    o_mine.step()
    assert torch.equal(mine[2], before[2]) and not torch.equal(mine[3], before[3])
    # more tensors than one launch carries
    many = [torch.nn.Parameter(torch.randn(5 + i, device=device)) for i in range(97)]
    many_ref = [torch.nn.Parameter(p.detach().clone()) for p in many]
    o_a, o_b = optim.Adam(many, lr=1e-2), torch.optim.Adam(many_ref, lr=1e-2)
    for _ in range(3):
        for p, q in zip(many, many_ref):
            p.grad = torch.randn_like(p); q.grad = p.grad.clone()
        o_a.step(); o_b.step()
    for p, q in zip(many, many_ref):
        assert_close(p, q, 2e-6, "adam many")
    assert float(o_a.state[many[0]]["step"]) == 3.0
    # misuse raises
    for bad in (torch.randn(3), torch.randn(3, dtype=torch.float64, device=device)):        # no CPU fallback, fp32 only
        q = torch.nn.Parameter(bad)
        q.grad = torch.zeros_like(q)
        with pytest.raises(ops.GlamHipError):
            optim.Adam([q]).step()
    with pytest.raises(ops.GlamHipError):
        optim.Adam(mine, amsgrad=True)


def test_adam_under_the_graphed_stepper_follows_the_eager_trajectory(device):
    """Training loop of trainer.py:286-301 with ``glam_amd.optim.Adam``: eager, one graph per batch, and 16 steps per graph launch give
    the same parameters bit for bit (device-side step count and learning rate), including a learning-rate change between epochs; and the
    trajectory agrees with ``torch.optim.Adam`` to rounding."""
    import copy
    from glam_amd import optim
    from glam_amd.data import DataLoader, synth_molecule
    from glam_amd.graphs import GraphedTrainStep
    rng = np.random.default_rng(4)
    mols = [synth_molecule(rng) for _ in range(24)]
    torch.manual_seed(6)
    net0 = model.Architecture(mol_block="_TripletMessage", message_steps=2, mol_readout="GlobalPool5", e_dim=64).to(device).train()   # RReLU, Dropout
    loss_fn = lambda out, b: torch.nn.functional.mse_loss(out.view(-1), b.y.view(-1))
    results = {}
    for mode in ("eager", "graph", "multi", "library"):
        net = copy.deepcopy(net0)
        ops.manual_seed(77, device)
        opt = (torch.optim.Adam(net.parameters(), lr=2.0 ** -9, capturable=True) if mode == "library"
               else optim.Adam(net.parameters(), lr=2.0 ** -9))
        loader = DataLoader(mols, batch_size=8, device=device)
        stepper = GraphedTrainStep(net, opt, loss_fn)
        for epoch in range(4):
            if epoch == 2:
                opt.param_groups[0]["lr"] = 2.0 ** -11
            if mode == "multi":
                stepper.run(list(loader), steps_per_graph=16)
            else:
                for b in loader:
                    if mode == "graph":
                        stepper(b)
                    else:
                        opt.zero_grad(set_to_none=True)
                        loss_fn(net(b), b).backward()
                        opt.step()
        results[mode] = [p.detach().clone() for p in net.parameters()]
    for a, b_, c in zip(results["eager"], results["graph"], results["multi"]):
        assert torch.equal(a, b_) and torch.equal(a, c)
    for a, r in zip(results["eager"], results["library"]):
        assert_close(a, r, 2e-5, "vs torch.optim.Adam")


def test_adam_kernel_against_the_oracle_restatement(device):
    """``k_adam`` through ``glam_amd.optim.Adam`` against ``oracle.adam_step`` (the same fp32 formula in numpy) on one ragged tensor
    list, 40 steps, device learning rate: 5e-7 of the largest parameter — the difference is the device's expm1 / sqrt / divide
    rounding, an order of magnitude below the distance of either to the library optimizer."""
    from glam_amd import optim
    from oracle import glam_oracle as oracle
    rng = np.random.default_rng(2)
    shapes = [(3,), (61, 7), (2050,)]
    ps = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    qs = [torch.nn.Parameter(torch.from_numpy(p.copy()).to(device)) for p in ps]
    lr_t = torch.tensor(2e-3, device=device)
    opt = optim.Adam(qs, lr=lr_t, weight_decay=1e-3)
    ms, vs = [np.zeros_like(p) for p in ps], [np.zeros_like(p) for p in ps]
    for step in range(40):
        for i, q in enumerate(qs):
            g = rng.standard_normal(shapes[i]).astype(np.float32)
            q.grad = torch.from_numpy(g).to(device)
            ps[i], ms[i], vs[i] = oracle.adam_step(ps[i], g, ms[i], vs[i], step, lr=float(np.float32(2e-3)), weight_decay=1e-3)
        opt.step()
    for p, q in zip(ps, qs):
        assert_close(q, torch.from_numpy(p), 5e-7, "k_adam vs oracle")


@pytest.mark.parametrize("n", [1, 33, 1024, 1025, 12288, 631808])
def test_loss_launch_matches_the_library_losses(device, n):
    """``glam_amd.loss`` (value + gradient in one launch, one scale launch backwards) against the criteria of the reference's trainers:
    nn.MSELoss (trainer.py:296), nn.BCEWithLogitsLoss (loss.py:48) and the classification loop's mean over the labels present
    (``criterion(y_score[y_true >= 0], y_true[y_true >= 0].float())``, trainer.py:244-245).  1e-6 relative on the value (fp32 sums in
    a different, fixed order), 1e-6 on the gradient; a non-unit upstream gradient; run-to-run bit-reproducible."""
    from glam_amd import loss
    torch.manual_seed(n)
    x0 = torch.randn(n, device=device) * 3
    y_reg = torch.randn(n, device=device)
    y_cls = torch.randint(-1, 2, (n,), device=device).float()
    if n > 1:
        y_cls[0] = 1.0
    cases = [("mse", lambda x: loss.get_loss("mse")(x, y_reg), lambda x: torch.nn.functional.mse_loss(x, y_reg)),
             ("bcel", lambda x: loss.get_loss("bcel")(x, y_cls.clamp(min=0)),
              lambda x: torch.nn.functional.binary_cross_entropy_with_logits(x, y_cls.clamp(min=0))),
             ("bcel_masked", lambda x: loss.get_loss("bcel_masked")(x, y_cls),
              lambda x: torch.nn.functional.binary_cross_entropy_with_logits(x[y_cls >= 0], y_cls[y_cls >= 0]))]
    for name, mine, ref in cases:
        if name == "bcel_masked" and not bool((y_cls >= 0).any()):
            continue
        xa, xb = x0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
        la, lb = mine(xa), ref(xb)
        (la * 0.37).backward(); (lb * 0.37).backward()
        va, vb = float(la.detach()), float(lb.detach())
        assert abs(va - vb) <= 1e-6 * max(1.0, abs(vb)), (name, va, vb)
        assert_close(xa.grad, xb.grad, 1e-6, f"{name} gradient")
        xc = x0.clone().requires_grad_(True)
        lc = mine(xc)
        (lc * 0.37).backward()
        assert torch.equal(lc, la) and torch.equal(xc.grad, xa.grad)          # fixed summation order: the same bits every run
    # shapes other than flat, and the empty selection of the masked mean (nan, as the reference's mean over nothing)
    if n == 1024:
        x2 = x0.view(32, 32).clone().requires_grad_(True)
        l2 = loss.mse_loss(x2, y_reg.view(32, 32))
        l2.backward()
        assert x2.grad.shape == (32, 32)
        assert torch.isnan(loss.bce_with_logits(x0, -torch.ones_like(x0), masked=True))
        with pytest.raises(ops.GlamHipError):
            loss.mse_loss(x0, y_reg[:-1])
        with pytest.raises(ops.GlamHipError):
            loss.mse_loss(x0.cpu(), y_reg.cpu())


def test_cached_staging_reuses_the_images_until_a_parameter_is_written(device):
    """``ops.cached_staging()``: the parameter re-layout launch (k_stage_params) runs once, not once per pass, gives the same output and
    gradients bit for bit, and runs again as soon as a parameter's version counter moves."""
    from glam_amd import _lib
    b = synth_batch(40, seed=3).to(device)
    torch.manual_seed(1)
    conv = layer.TripletMessage(60, 4).to(device)
    x = torch.randn(b.x.size(0), 60, device=device, requires_grad=True)

    def run():
        with _lib.kernel_timer(capacity=64) as kt:
            out = conv(x, b.edge_index, b.edge_attr)
            g = torch.autograd.grad(out.sum(), [x] + list(conv.parameters()))
        return out, g, sum("k_stage_params" in n for n, _, _ in kt.records())

    ref = run()
    assert ref[2] == 1
    with ops.cached_staging():
        first, second = run(), run()
        assert first[2] == 1 and second[2] == 0
        for got in (first, second):
            assert torch.equal(got[0], ref[0]) and all(torch.equal(a, c) for a, c in zip(got[1], ref[1]))
        with torch.no_grad():
            conv.weight_scale.mul_(0.5)          # an in-place write: the images are stale
        third = run()
        assert third[2] == 1 and not torch.equal(third[0], ref[0])
        assert run()[2] == 0
    fresh = run()                                 # outside the context: per-pass staging again, same numbers as the cached images gave
    assert fresh[2] == 1 and torch.equal(fresh[0], third[0])


def test_cached_staging_sees_the_library_optimizer_and_never_caches_inside_a_capture(device):
    """``glam_amd.optim.Adam`` writes parameters through raw device pointers (no version counter moves), so its ``step`` bumps
    ``ops.PARAM_EPOCH`` and the staged images are rebuilt: a training loop inside ``ops.cached_staging()`` follows the same trajectory as
    one with per-pass staging, bit for bit.  And images built while a stream capture is running (graph-pool memory, filled only on
    replay) never enter the cache: an eager pass right after the capture stages its own."""
    import copy
    from glam_amd import _lib, optim
    b = synth_batch(24, seed=5).to(device)
    torch.manual_seed(2)
    conv0 = layer.TripletMessage(60, 4).to(device)
    x = torch.randn(b.x.size(0), 60, device=device)

    def train(cached, steps=4):
        conv = copy.deepcopy(conv0)
        opt = optim.Adam(conv.parameters(), lr=1e-2)
        outs, staged = [], 0
        with ops.cached_staging(cached):
            for _ in range(steps):
                with _lib.kernel_timer(capacity=64) as kt:
                    out = conv(x, b.edge_index, b.edge_attr)
                    opt.zero_grad(set_to_none=True)
                    (out * out).sum().backward()
                staged += sum("k_stage_params" in n for n, _, _ in kt.records())
                opt.step()
                outs.append(out.detach().clone())
        return outs, staged

    ref, n_ref = train(False)
    got, n_got = train(True)
    assert n_ref == 4 and n_got == 4                  # every step follows an optimizer write: every step re-stages
    assert not torch.equal(ref[0], ref[1])            # (the parameters did move)
    for a, c in zip(got, ref):
        assert torch.equal(a, c)

    # a cache miss inside a capture is not published
    conv = copy.deepcopy(conv0)
    with ops.cached_staging():
        ops._STAGED.clear()
        ops.graph_index(b.edge_index, b.x.size(0)).ell(); ops.graph_index(b.edge_index, b.x.size(0)).ell_t()
        with torch.no_grad():
            warm = conv(x, b.edge_index, b.edge_attr)         # eager visit: read-backs done, cache filled ...
        ops._STAGED.clear()                                   # ... and emptied again: the capture below misses
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side), torch.no_grad():
            with torch.cuda.graph(g, stream=side):
                cap = conv(x, b.edge_index, b.edge_attr)
        torch.cuda.current_stream().wait_stream(side)
        assert not ops._STAGED                                # nothing published from inside the capture
        with torch.no_grad():
            eager = conv(x, b.edge_index, b.edge_attr)        # BEFORE the first replay: must not read the graph pool's images
        assert torch.equal(eager, warm)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(cap, warm)


def test_adam_device_step_count_equals_the_number_of_replays(device):
    """The launch reads the device step count and learning rate with agent-scope atomic loads (a plain load could be served a line
    cached before the previous launch's update: stale bias correction, a stalled counter): after one eager step and k replays of a
    captured one the count is k + 1 and the parameters equal k + 1 eager steps bit for bit; ``state_dict()`` hands out a private
    ``step`` per parameter; a deep copy of the optimizer owns its moments."""
    import copy
    from glam_amd import optim
    torch.manual_seed(0)
    shapes = (5, 1024, 4099, 70000)
    init = [torch.randn(n, device=device) for n in shapes]
    grads = [torch.randn(n, device=device) for n in shapes]

    def make():
        ps = [torch.nn.Parameter(t.clone()) for t in init]
        for p, g in zip(ps, grads):
            p.grad = g.clone()
        return ps, optim.Adam(ps, lr=1e-2)

    K = 37
    ps_e, opt_e = make()
    for _ in range(K + 1):
        opt_e.step()
    ps_g, opt_g = make()
    opt_g.step()                                   # builds the plan eagerly (allocations, address table)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        opt_g.step()
    for _ in range(K):
        gr.replay()
    torch.cuda.synchronize()
    assert float(opt_g.state[ps_g[0]]["step"]) == K + 1 == float(opt_e.state[ps_e[0]]["step"])
    for a, b in zip(ps_e, ps_g):
        assert torch.equal(a, b)
    sd = opt_g.state_dict()
    steps = [st["step"] for st in sd["state"].values()]
    assert len({t.data_ptr() for t in steps}) == len(steps) and all(float(t) == K + 1 for t in steps)
    twin = copy.deepcopy(opt_g)
    m0 = opt_g.state[ps_g[1]]["exp_avg"].clone()
    twin.step()                                    # must write the COPY's moments and parameters, not the original's
    torch.cuda.synchronize()
    assert torch.equal(opt_g.state[ps_g[1]]["exp_avg"], m0) and torch.equal(ps_g[1], ps_e[1])


def test_loss_accepts_integer_labels_like_the_reference_call_site(device):
    """``criterion(y_score, y_true)`` with the LongTensor labels of src_1gp/dataset.py:139 (the reference writes ``.float()`` at the call
    site, trainer.py:244-245): the masked BCE casts inside and equals the float call."""
    from glam_amd import loss
    g = torch.Generator().manual_seed(3)
    score = torch.randn(64, 12, generator=g).to(device).requires_grad_(True)
    y = torch.randint(-1, 2, (64, 12), generator=g).to(device)
    crit = loss.get_loss("bcel_masked")
    a = crit(score, y)
    b = crit(score, y.float())
    assert torch.equal(a, b)
    (ga,), (gb,) = torch.autograd.grad(a, [score]), torch.autograd.grad(b, [score])
    assert torch.equal(ga, gb)


def test_ts_gemm_pair_with_different_depths_sizes_its_image_for_the_larger(device):
    """Two products of one kernel variant but different K in one launch (glam_ts_gemm_pair): the LDS allocation must hold the larger
    image (it was sized from the first job only)."""
    from glam_amd import _lib
    lib, p = _lib.load(), _lib.ptr
    g = torch.Generator().manual_seed(8)
    N, M = 500, 180
    outs, refs, args = [], [], []
    for K in (16, 60):
        A = torch.randn(N, K, generator=g).to(device)
        W = (torch.randn(K, M, generator=g) / K ** 0.5).to(device)
        img = torch.empty(lib.glam_ts_gemm_image_bytes(K, M) // 4, device=device)
        assert lib.glam_ts_gemm_make_image(p(W), M, 0, K, M, p(img), _lib.stream()) == 0
        out = torch.empty(N, M, device=device)
        outs.append(out); refs.append((A.double() @ W.double()).float()); args.append((A, K, img, out))
    (Aa, Ka, ia, oa), (Ab, Kb, ib, ob) = args
    rc = lib.glam_ts_gemm_pair(p(Aa), Ka, Ka, 0, p(ia), None, p(oa), M, M, None, 0, None, 0,
                               p(Ab), Kb, Kb, 0, p(ib), None, p(ob), M, M, None, 0, None, 0, N, _lib.stream())
    assert rc == 0, lib.glam_last_error()
    torch.cuda.synchronize()
    for o, r in zip(outs, refs):
        assert_close(o, r, 1e-5, "ts_gemm pair, K = 16 | 60")


@pytest.mark.parametrize("De,Hd,C", [(4, 32, 60), (8, 32, 45), (3, 64, 15), (1, 4, 7), (5, 16, 33), (3, 17, 15)])
def test_relation_mlp_against_torch_and_fp64(device, De, Hd, C):
    """``ops.relation_mlp`` (csrc/relmlp.hip: nn(eye(De)) of NNConv's edge network, src_1gp/layer.py:115-122) against the same modules run
    by torch in fp64: the table and the four parameter gradients within the fp64-twin bound."""
    torch.manual_seed(De * 100 + C)
    nn_ = torch.nn.Sequential(torch.nn.Linear(De, Hd), torch.nn.ReLU(), torch.nn.Linear(Hd, C * C))
    cot = torch.randn(De, C * C)
    res = {}
    for dt in (torch.float32, torch.float64):
        import copy
        m = copy.deepcopy(nn_).to(dt)
        o = m(torch.eye(De, dtype=dt))
        res[dt] = (o.detach(), torch.autograd.grad((o * cot.to(dt)).sum(), list(m.parameters())))
    nn_ = nn_.to(device)
    out = ops.relation_mlp(nn_, De)
    gs = torch.autograd.grad((out * cot.to(device)).sum(), list(nn_.parameters()))
    assert_fp32_parity(out, res[torch.float64][0], res[torch.float32][0], "relation table", out_tol=1e-5)
    for n, a, r64, r32 in zip(["w1", "b1", "w2", "b2"], gs, res[torch.float64][1], res[torch.float32][1]):
        assert_fp32_parity(a, r64, r32, "relation mlp d_" + n)


@pytest.mark.parametrize("a_kc,b_kc", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("R,Cn,K", [(1024, 1024, 300), (100, 300, 1024), (257, 75, 33), (5, 617, 450)])
def test_dense_gemm_c_abi(device, a_kc, b_kc, R, Cn, K):
    """glam_dense_gemm in all four layout combinations, with and without gate / bias / activation / the all-ones column, 16-byte and
    scalar access paths, ragged tiles and a partial last chunk, against the fp64 product: fp32 accuracy from the 3 x bf16 form."""
    lib, p, st = ops._lib.load(), ops._lib.ptr, ops._lib.stream
    g = torch.Generator().manual_seed(R * 7 + Cn * 3 + K + 2 * a_kc + b_kc)
    A, G, B, bv = (torch.randn(R, K, generator=g), torch.randn(R, K, generator=g), torch.randn(K, Cn, generator=g),
                   torch.randn(Cn, generator=g))
    Ad, Gd = ((A, G) if a_kc else (A.t().contiguous(), G.t().contiguous()))
    Ad, Gd, Bd, bd = Ad.to(device), Gd.to(device), (B.t().contiguous() if b_kc else B).to(device), bv.to(device)
    for gate, bias, act, ones in [(False, False, 0, False), (True, True, 1, True), (True, False, 2, False), (False, True, 2, True)]:
        ones = ones and not b_kc and Cn > 1
        gs, sl = 0.25, 0.125
        Ag = A.double() * torch.where(G > 0, 1.0, gs).double() if gate else A.double()
        ref = Ag @ B.double() + (bv.double() if bias else 0.0)
        ref = ref.clamp_min(0) if act == 1 else torch.where(ref > 0, ref, ref * sl) if act == 2 else ref
        ldc = Cn + 4
        C = torch.full((R, ldc), float("nan"), device=device)
        rs = torch.full((R,), float("nan"), device=device)
        rc = lib.glam_dense_gemm(p(Ad), K if a_kc else 1, 1 if a_kc else R, p(Gd) if gate else None, gs, p(Bd), 1 if b_kc else Cn,
                                 K if b_kc else 1, p(bd) if bias else None, act, sl, p(C), ldc, p(rs) if ones else None, R, Cn, K, st())
        assert rc == 0, lib.glam_last_error()
        tol = 2e-6 * max(1.0, K ** 0.5 / 8)
        err = (C[:, :Cn].cpu().double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        assert err < tol, (gate, bias, act, ones, err)
        assert torch.isnan(C[:, Cn:]).all(), "wrote beyond Cn"
        if ones:
            want = Ag.sum(1)
            assert (rs.cpu().double() - want).abs().max().item() / max(1.0, want.abs().max().item()) < tol


@pytest.mark.parametrize("N,K,M,act", [(1024, 300, 1024, "relu"), (642, 300, 1024, "leaky"), (33, 64, 128, "none"), (2039, 300, 1024, "relu"),
                                        (1024, 75, 1024, "relu"), (257, 450, 1024, "leaky"), (64, 1024, 617, "none")])
def test_linear_dense_forward_backward_against_fp64(device, N, K, M, act):
    """The readout MLP's linear on the dense kernel (ops.linear_act -> glam_linear_dense_fwd / _bwd: bias + activation in the epilogue,
    activation derivative + dx + dw + db in one launch) against F.linear + the activation in fp64 (src_1gp/model.py:43-45, 60)."""
    import torch.nn.functional as F
    torch.manual_seed(N + K)
    x = torch.randn(N, K, device=device, requires_grad=True)
    w = (torch.randn(M, K, device=device) * K ** -0.5).requires_grad_(True)
    b = torch.randn(M, device=device, requires_grad=True)
    code = {"none": 0, "relu": 1, "leaky": 2}[act]
    y = ops.linear_act(x, w, b, act, 0.2) if K % 4 == 0 and M % 4 == 0 else ops._LinearDense.apply(x, w, b, code, 0.2)
    assert y is not None      # (shapes with partial quads are not routed here by ops.linear — slower than the library — but are correct)
    cot = torch.randn_like(y)
    got = torch.autograd.grad(y, (x, w, b), cot)
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    pre = F.linear(xd, wd, bd)
    # the derivative follows the sign of the kernel's own output (an element whose pre-activation is within rounding of zero may fall on
    # either side of the kink: of 2 M elements a few do)
    slope = {"relu": 0.0, "leaky": 0.2, "none": 1.0}[act]
    yd = pre * torch.where(y.detach().double() > 0, 1.0, slope)
    ref = torch.autograd.grad(yd, (xd, wd, bd), cot.double())
    want = torch.relu(pre) if act == "relu" else F.leaky_relu(pre, 0.2) if act == "leaky" else pre
    assert ((y.double() - want).abs().max() / want.abs().max()).item() < 2e-6
    for gg, rr, name in zip(got, ref, ("dx", "dw", "db")):
        assert ((gg.double() - rr).abs().max() / rr.abs().max()).item() < 3e-6 * max(1.0, N ** 0.5 / 16), name
    # without input gradient (the first layer of a model) and without bias
    y2 = ops._LinearDense.apply(x.detach(), w, None, code, 0.2)
    gw, = torch.autograd.grad(y2, (w,), cot)
    yd2 = F.linear(xd.detach(), wd) * torch.where(y2.detach().double() > 0, 1.0, slope)
    rw, = torch.autograd.grad(yd2, (wd,), cot.double())
    assert ((gw.double() - rw).abs().max() / rw.abs().max()).item() < 3e-6 * max(1.0, N ** 0.5 / 16)


@pytest.mark.parametrize("N,K,M,act", [(32, 300, 1024, 1), (8, 300, 1024, 2), (32, 1024, 300, 0), (64, 300, 1024, 1), (33, 64, 128, 0), (4, 32, 4, 1),
                                        (100, 1024, 617, 2), (1024, 300, 1024, 1)])
def test_linear_dense_k_split_across_blocks(device, N, K, M, act):
    """glam_linear_dense_fwd_ws / _bwd_ws (the readout MLP at the reference's batch of 32, run.py:40: few tiles, long reductions — k split
    across blocks, partial tiles added in split order by a second launch): against fp64; the same bits on every run and whatever the
    workspace held before; with many tiles (N = 1024) and with ws = NULL exactly the plain entry points."""
    import torch.nn.functional as F
    lib, p, st = ops._lib.load(), ops._lib.ptr, ops._lib.stream
    torch.manual_seed(N * 7 + K)
    x, w, b = torch.randn(N, K, device=device), torch.randn(M, K, device=device) * K ** -0.5, torch.randn(M, device=device)
    dy = torch.randn(N, M, device=device)
    nb = lib.glam_dense_ws_bytes()
    ws = torch.full((nb,), 0xFF, dtype=torch.uint8, device=device)          # (NaN patterns: nothing of it may reach a result)
    nan = lambda *s: torch.full(s, float("nan"), device=device)
    slope = 0.2
    def run(ws_t):
        y, dx, dw, db = nan(N, M), nan(N, K), nan(M, K), nan(M)
        a = (p(ws_t), nb) if ws_t is not None else (None, 0)
        assert lib.glam_linear_dense_fwd_ws(p(x), p(w), p(b), N, K, M, act, slope, p(y), *a, st()) == 0, lib.glam_last_error()
        gate = p(y) if act else None
        if N >= 4:
            assert lib.glam_linear_dense_bwd_ws(p(x), p(w), p(dy), gate, 0.0 if act == 1 else slope, N, K, M, p(dx), p(dw), p(db), *a, st()) == 0, \
                lib.glam_last_error()
        return y, dx, dw, db
    got = run(ws)
    for _ in range(3):
        again = run(ws)
        for u, v in zip(got, again):
            assert torch.equal(u, v)
    plain = run(None)
    yp, dxp, dwp, dbp = nan(N, M), nan(N, K), nan(M, K), nan(M)
    assert lib.glam_linear_dense_fwd(p(x), p(w), p(b), N, K, M, act, slope, p(yp), st()) == 0
    assert torch.equal(plain[0], yp)
    if N >= 1024:
        for u, v in zip(got, plain):
            assert torch.equal(u, v)                                                           # many tiles: nothing is split
    xd, wd, bd = x.double(), w.double(), b.double()
    pre = F.linear(xd, wd, bd)
    want = torch.relu(pre) if act == 1 else F.leaky_relu(pre, slope) if act == 2 else pre
    assert ((got[0].double() - want).abs().max() / want.abs().max()).item() < 2e-6
    assert not torch.isnan(got[0]).any()
    g = dy.double() * (torch.where(got[0].double() > 0, 1.0, 0.0 if act == 1 else slope) if act else 1.0)
    tol = 3e-6 * max(1.0, N ** 0.5 / 16)
    for name, u, r in (("dx", got[1], g @ wd), ("dw", got[2], g.t() @ xd), ("db", got[3], g.sum(0))):
        assert ((u.double() - r).abs().max() / r.abs().max()).item() < tol, name
    # a workspace that is too small or misaligned is refused
    assert lib.glam_linear_dense_fwd_ws(p(x), p(w), p(b), N, K, M, act, slope, p(got[0]), p(ws), nb - 16, st()) == ops._lib.GLAM_E_INVALID
    assert lib.glam_linear_dense_fwd_ws(p(x), p(w), p(b), N, K, M, act, slope, p(got[0]), ws.data_ptr() + 4, nb, st()) == ops._lib.GLAM_E_INVALID


@pytest.mark.parametrize("N,K,M,diff_x", [(20400, 15, 60, False), (606, 15, 60, False), (777, 60, 60, True), (64, 32, 44, True), (5, 15, 60, False)])
def test_linear_block_relu_in_the_tall_products_epilogue(device, N, K, M, diff_x):
    """LinearBlock(K, M, act=ReLU) where the product runs on k_tall_x3 (the input embedding of the models, src_1gp/model.py:40 /
    layer.py:232-237): glam_ts_gemm_relu applies the activation in the epilogue — the same values as linear + relu, bit for bit, and
    the same gradients (the backward masks dy by the saved output)."""
    torch.manual_seed(N + K)
    blk = layer.LinearBlock(K, M, act="ReLU()").to(device)
    x = torch.randn(N, K, device=device, requires_grad=diff_x)
    y = blk(x)
    lin = ops.linear(x, blk.linear.weight, blk.linear.bias)
    ref = torch.relu(lin)
    fused = type(y.grad_fn).__name__.startswith("_Linear") and not type(y.grad_fn).__name__.startswith("_LinearDense")
    assert fused == bool(ops._lib.load().glam_ts_gemm_relu_supported((K + 3) // 4 * 4, M)) or (K % 4 and diff_x)
    assert torch.equal(y, ref)
    cot = torch.randn_like(y)
    wrt = [blk.linear.weight, blk.linear.bias] + ([x] if diff_x else [])
    g1 = torch.autograd.grad(y, wrt, cot)
    g0 = torch.autograd.grad(ref, wrt, cot)
    for u, v in zip(g1, g0):
        assert torch.equal(u, v)
    xd, wd, bd = x.detach().double(), blk.linear.weight.detach().double(), blk.linear.bias.detach().double()
    assert_close(y, torch.relu(torch.nn.functional.linear(xd, wd, bd)), 2e-6, "linear + relu")
    lib = ops._lib.load()
    assert lib.glam_ts_gemm_relu_supported(300, 1024) == 0 and lib.glam_ts_gemm_relu_supported(15, 60) == 0


@pytest.mark.parametrize("N", [1, 37, 606, 20400, 40000])
def test_relu_backward_inside_the_weight_gradient_product(device, N, monkeypatch):
    """The first linear of the models (15 -> 60 + ReLU, src_1gp/model.py:40; atom features need no gradient): the ReLU's backward
    ``dy * (y > 0)`` runs inside the weight-gradient product (glam_wgrad_gemm_split_relu: one launch less per step) — the gradients of the
    elementwise-launch route bit for bit, at sizes on both sides of the row counts where the product changes its grid."""
    from glam_amd import _lib
    torch.manual_seed(N)
    blk = layer.LinearBlock(15, 60, act="ReLU()").to(device)
    x = torch.randn(N, 15, device=device)
    cot = torch.randn(N, 60, device=device)
    res = {}
    fused = bool(_lib.load().glam_ts_gemm_relu_supported(16, 60))      # (GLAM_X3=0: no ReLU epilogue, so no mask to fold either)
    monkeypatch.delenv("GLAM_WGRAD_X3_ROWS", raising=False)
    for inside in (True, False):
        monkeypatch.setattr(ops, "RELU_IN_WGRAD", inside)
        y = blk(x)
        with _lib.kernel_timer(capacity=16) as kt:
            res[inside] = torch.autograd.grad(y, [blk.linear.weight, blk.linear.bias], cot)
        names = [n for n, _, _ in kt.records()]
        assert any("relu mask" in n for n in names) == (inside and fused), names
    for u, v in zip(res[True], res[False]):
        if N < 32768:
            assert torch.equal(u, v)
        else:       # (from 32 768 rows the unmasked product runs on k_wgrad_x3, the masked one stays on k_wgrad: other roundings)
            assert_close(u, v, 3e-6 * N ** 0.5 / 10, "masked vs unmasked route")
    yd = torch.relu(torch.nn.functional.linear(x.double(), blk.linear.weight.double(), blk.linear.bias.double()))
    gd = cot.double() * (yd > 0)
    assert_close(res[True][0], gd.t() @ x.double(), 3e-6 * max(1.0, N ** 0.5 / 10), "d_weight")
    assert_close(res[True][1], gd.sum(0), 3e-6 * max(1.0, N ** 0.5 / 10), "d_bias")


def test_linear_block_routes_the_readout_mlp_to_the_dense_kernel(device):
    """LinearBlock(300, 1024, act=ReLU) — `mol_flat` of the parity configuration — runs as one dense launch each way and agrees with
    the unfused composition; shapes outside the class (K = 450: rows not 16-byte multiples) keep the library route."""
    torch.manual_seed(3)
    blk = layer.LinearBlock(300, 1024, act="ReLU()").to(device)
    x = torch.randn(256, 300, device=device, requires_grad=True)
    y = blk(x)
    assert type(y.grad_fn).__name__.startswith("_LinearDense")
    ref = torch.relu(torch.nn.functional.linear(x, blk.linear.weight, blk.linear.bias))
    assert_close(y, ref, 5e-6, "LinearBlock dense")
    gy, gr = torch.autograd.grad(y.sum() + (y * y).sum(), x)[0], torch.autograd.grad(ref.sum() + (ref * ref).sum(), x)[0]
    assert_close(gy, gr, 1e-5, "LinearBlock dense dx")
    # rows that are not multiples of 16 bytes (hid_dim 90: 450 columns) stay on the library (measured slower on the dense kernel)
    blk2 = layer.LinearBlock(450, 1024, act="ReLU()").to(device)
    assert not type(blk2(torch.randn(64, 450, device=device)).grad_fn).__name__.startswith("_LinearDense")


def test_dense_gemm_rounding_is_unbiased(device):
    """The bf16 matrix instruction aligns its products and the accumulator by truncation: with ONE accumulator per tile across a long
    reduction the 3 x bf16 product carries a bias (measured: mean error -0.1 rms, z = -80 over 6e5 elements, coherent in every sum
    taken downstream).  The kernel keeps the small / middle / large partial products in separate chains; this pins the property:
    the mean error of a K = 1024 product is zero within its sampling noise, and its rms is below the fp32 library product's."""
    torch.manual_seed(0)
    N, K, M = 2039, 300, 1024
    w = ((torch.rand(M, K, device=device) * 2 - 1) * K ** -0.5)
    dy, x = torch.rand(N, M, device=device), torch.randn(N, K, device=device)        # positive dy: every product term has the same sign
    dx, dw, db = torch.empty(N, K, device=device), torch.empty(M, K, device=device), torch.empty(M, device=device)
    lib, p = ops._lib.load(), ops._lib.ptr
    assert lib.glam_linear_dense_bwd(p(x), p(w), p(dy), None, 0.0, N, K, M, p(dx), p(dw), p(db), ops._lib.stream()) == 0
    ref = dy.double() @ w.double()
    e = dx.double() - ref
    rms = e.pow(2).mean().sqrt().item()
    z = e.mean().item() / rms * e.numel() ** 0.5
    e_lib = ((dy @ w).double() - ref).pow(2).mean().sqrt().item()
    assert abs(z) < 6.0, f"mean error {e.mean().item():+.2e} is {z:+.1f} sigma from zero (rms {rms:.2e})"
    assert rms < e_lib, f"rms error {rms:.2e} vs the library's {e_lib:.2e}"


def test_prestage_builds_the_pass_weights_in_one_launch(device, monkeypatch):
    """ops.prestage / glam_prestage: the input linear's GEMM image, the TripletMessage's staged images and the GRU's gate matrices (the
    pre-split images of its warp-specialised step; GLAM_GRU_PRE=0: the four k_ts_gemm images) come from ONE launch at the head of a
    model pass (three launches before), land in the weight scope under the keys the ops look them up with (no k_stage_params /
    k_ts_make_image(s) / k_gru_ws_pre launch follows), and the pass is bit-identical to the lazily staged one."""
    from glam_amd._lib import kernel_timer
    torch.manual_seed(11)
    b = synth_batch(48, seed=5).to(device)
    net = model.Architecture(mol_block="_TripletMessage", message_steps=3, pre_act="ReLU", graph_act="ReLU", flat_act="ReLU",
                             graph_do="_None()", end_do="_None()").to(device)
    monkeypatch.setattr(ops, "USE_TORCH_EXT", False)       # (the eager C++ node stages per call; a captured step uses this route)

    def run(flag, pre=True):
        monkeypatch.setattr(ops, "PRESTAGE", flag)
        monkeypatch.setattr(ops, "GRU_PRE", pre)
        net.zero_grad()
        with kernel_timer() as kt:
            out = net(b)
            out.square().sum().backward()
        return out.detach().clone(), [p.grad.clone() for p in net.parameters()], [r[0] for r in kt.records()]

    o1, g1, k1 = run(True)
    o0, g0, k0 = run(False)
    assert torch.equal(o1, o0)
    for a, c in zip(g1, g0):
        assert torch.equal(a, c)
    staging = lambda names: [n for n in names if "k_stage_params" == n.split("<")[0] or "k_ts_make_image" in n or "k_prestage" in n
                             or "k_gru_ws_pre" in n]
    assert staging(k1) == ["k_prestage"], staging(k1)
    pre_route = ops.gru_images_pre(int(b.x.size(0)), net.mol_conv.gru.weight_ih_l0.size(1))      # (off with GLAM_X3=0: the GRU step is not warp-specialised)
    assert len(staging(k0)) == 3 and "k_prestage" not in staging(k0) and ("k_gru_ws_pre" in staging(k0)) == pre_route, staging(k0)
    o2, g2, k2 = run(True, pre=False)               # the plain images of the gate matrices: the same pass, bit for bit
    o3, g3, k3 = run(False, pre=False)
    assert torch.equal(o2, o1) and torch.equal(o3, o1)
    for a, c, d in zip(g1, g2, g3):
        assert torch.equal(a, c) and torch.equal(a, d)
    assert staging(k2) == ["k_prestage"] and len(staging(k3)) == 3 and "k_gru_ws_pre" not in staging(k3), (staging(k2), staging(k3))
    # C ABI: the image table is bounded, empty calls and shapes outside the kernel table are refused
    lib = ops._lib.load()
    import ctypes
    assert lib.glam_prestage(*([None] * 5 + [0] * 5 + [None]), 0, None, None, None, None) == ops._lib.GLAM_E_INVALID
    w = torch.randn(60, 60, device=device)
    img = torch.empty(lib.glam_ts_gemm_image_bytes(60, 60) // 4, device=device)
    vp = ctypes.c_void_p
    assert lib.glam_prestage(*([None] * 5 + [0] * 5 + [None]), 7, (vp * 7)(*[w.data_ptr()] * 7), (ctypes.c_int32 * 28)(*([60, 1, 60, 60] * 7)),
                             (vp * 7)(*[img.data_ptr()] * 7), ops._lib.stream()) == ops._lib.GLAM_E_INVALID
    assert lib.glam_prestage(*([None] * 5 + [0] * 5 + [None]), 1, (vp * 1)(w.data_ptr()), (ctypes.c_int32 * 4)(60, 1, 500, 500),
                             (vp * 1)(img.data_ptr()), ops._lib.stream()) == ops._lib.GLAM_E_UNSUPPORTED
    ref = torch.empty_like(img)
    assert lib.glam_ts_gemm_make_image(w.data_ptr(), 60, 1, 60, 60, ref.data_ptr(), ops._lib.stream()) == 0
    assert lib.glam_prestage(*([None] * 5 + [0] * 5 + [None]), 1, (vp * 1)(w.data_ptr()), (ctypes.c_int32 * 4)(60, 1, 60, 60),
                             (vp * 1)(img.data_ptr()), ops._lib.stream()) == 0
    assert torch.equal(img, ref)
    # the pre-split images of a GRU's gate matrices as four jobs (transW = 2 / 3) = glam_gru_ws_make_pre
    C = 44
    w_ih, w_hh = torch.randn(3 * C, C, device=device), torch.randn(3 * C, C, device=device)
    nb = lib.glam_gru_ws_pre_bytes()
    want = torch.full((2, nb), 0xCD, dtype=torch.uint8, device=device)
    got = torch.full((2, nb), 0xCD, dtype=torch.uint8, device=device)
    assert lib.glam_gru_ws_make_pre(w_ih.data_ptr(), w_hh.data_ptr(), C, want[0].data_ptr(), want[1].data_ptr(), ops._lib.stream()) == 0
    Wp = (vp * 4)(w_ih.data_ptr(), w_hh.data_ptr(), w_ih.data_ptr(), w_hh.data_ptr())
    ip = (vp * 4)(got[0].data_ptr(), got[0].data_ptr(), got[1].data_ptr(), got[1].data_ptr())
    assert lib.glam_prestage(*([None] * 5 + [0] * 5 + [None]), 4, Wp, (ctypes.c_int32 * 16)(C, 2, C, 0, C, 2, C, 1, C, 3, C, 0, C, 3, C, 1), ip,
                             ops._lib.stream()) == 0, lib.glam_last_error()
    assert torch.equal(got, want) and not (want == 0xCD).all()
    assert lib.glam_prestage(*([None] * 5 + [0] * 5 + [None]), 1, Wp, (ctypes.c_int32 * 4)(C, 2, C, 2), ip, ops._lib.stream()) == ops._lib.GLAM_E_INVALID
    assert lib.glam_prestage(*([None] * 5 + [0] * 5 + [None]), 1, Wp, (ctypes.c_int32 * 4)(20, 2, 20, 0), ip, ops._lib.stream()) == ops._lib.GLAM_E_UNSUPPORTED


def test_wgrad_split_writes_a_narrow_weight_gradient_contiguously(device, wgrad_route):
    """glam_wgrad_gemm_split with J = 15 over rows of 16 floats (the input linear 15 -> 60 on zero-padded atom features): dw comes out
    as a contiguous [I, 15] tensor (no strided view for autograd to copy), db beside it; nothing is written beyond them."""
    lib, p = ops._lib.load(), ops._lib.ptr
    torch.manual_seed(2)
    N, I, J = 5000, 60, 15
    dy = torch.randn(N, I, device=device)
    x = torch.randn(N, 16, device=device)
    x[:, 15] = 7.0                                            # (the pad column is not trusted to be zero)
    buf = torch.full((I * J + 8,), float("nan"), device=device)
    db = torch.full((I + 4,), float("nan"), device=device)
    ws = torch.empty(lib.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=device)
    assert lib.glam_wgrad_gemm_split(p(dy), I, I, p(x), J, 16, p(buf), p(db), N, p(ws), ws.numel(), ops._lib.stream()) == 0, lib.glam_last_error()
    ref = dy.double().t() @ x[:, :J].double()
    assert_close(buf[:I * J].view(I, J), ref, 2e-5, "narrow dw")
    assert_close(db[:I], dy.double().sum(0), 2e-5, "narrow db")
    assert torch.isnan(buf[I * J:]).all() and torch.isnan(db[I:]).all()
    assert lib.glam_wgrad_gemm_split(p(dy), I, I, p(x), J, 15, p(buf), p(db), N, p(ws), ws.numel(), ops._lib.stream()) != 0     # ldq < ceil4(J)
    lin = layer.LinearBlock(15, 60, act="_None").to(device)
    xin = torch.randn(300, 15, device=device)
    lin(xin).sum().backward()
    assert lin.linear.weight.grad.is_contiguous() and lin.linear.weight.grad.shape == (60, 15)
    assert_close(lin.linear.weight.grad, xin.sum(0).expand(60, 15), 2e-5, "LinearBlock(15, 60) weight gradient")


@pytest.mark.parametrize("nseg,N", [(1, 5000), (2, 2500), (3, 20400), (3, 777)])
def test_wgrad_pair_split_seg_c_abi(device, nseg, N, wgrad_route):
    """glam_wgrad_gemm_pair_split_seg: both weight-gradient products of a GRU summed over the operand sets of up to three applications
    in one launch + one reduction (waves whose row range straddles a set boundary included), with a carry addend, against fp64; one
    set is the plain entry point bit for bit; sets too short for a wave's row range are refused."""
    import ctypes
    lib, p = ops._lib.load(), ops._lib.ptr
    torch.manual_seed(nseg * 13 + N)
    C, M = 60, 180
    sets = [(torch.randn(N, M, device=device), torch.randn(N, C, device=device), torch.randn(N, M, device=device),
             torch.randn(N, C, device=device)) for _ in range(nseg)]
    add = [torch.randn(M, C, device=device), torch.randn(M, device=device), torch.randn(M, C, device=device), torch.randn(M, device=device)]
    out = [torch.full_like(t, float("nan")) for t in add]
    ws = torch.empty(lib.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=device)
    arr = lambda i: (ctypes.c_void_p * nseg)(*[t[i].data_ptr() for t in sets])
    rc = lib.glam_wgrad_gemm_pair_split_seg(nseg, arr(0), M, M, arr(1), C, C, 1, p(out[0]), p(out[1]), arr(2), M, M, arr(3), C, C, 0, p(out[2]),
                                            p(out[3]), N, p(ws), ws.numel(), p(add[0]), p(add[1]), p(add[2]), p(add[3]), ops._lib.stream())
    assert rc == 0, lib.glam_last_error()
    celu = lambda t: torch.nn.functional.celu(t.double())
    ref = [sum(t[0].double().t() @ celu(t[1]) for t in sets) + add[0].double(), sum(t[0].double().sum(0) for t in sets) + add[1].double(),
           sum(t[2].double().t() @ t[3].double() for t in sets) + add[2].double(), sum(t[2].double().sum(0) for t in sets) + add[3].double()]
    for o, r, name in zip(out, ref, ("dw_ih", "db_ih", "dw_hh", "db_hh")):
        assert_close(o, r, 3e-6 * max(1.0, (nseg * N) ** 0.5 / 16), name)
    if nseg == 1:
        one = [torch.empty_like(t) for t in add]
        t = sets[0]
        assert lib.glam_wgrad_gemm_pair_split(p(t[0]), M, M, p(t[1]), C, C, 1, p(one[0]), p(one[1]), p(t[2]), M, M, p(t[3]), C, C, 0, p(one[2]),
                                              p(one[3]), N, p(ws), ws.numel(), p(add[0]), p(add[1]), p(add[2]), p(add[3]), ops._lib.stream()) == 0
        for a, b in zip(out, one):
            assert torch.equal(a, b)
    short = [(torch.randn(2, M, device=device), torch.randn(2, C, device=device)) * 2 for _ in range(3)]
    arr2 = lambda i: (ctypes.c_void_p * 3)(*[t[i].data_ptr() for t in short])
    rc = lib.glam_wgrad_gemm_pair_split_seg(3, arr2(0), M, M, arr2(1), C, C, 0, p(out[0]), p(out[1]), arr2(2), M, M, arr2(3), C, C, 0, p(out[2]),
                                            p(out[3]), 2, p(ws), ws.numel(), None, None, None, None, ops._lib.stream())
    if wgrad_route == "x3" and ops._lib.route_enabled("wgrad_x3"):      # k_wgrad_x3 deals its blocks set by set: any set length runs
        assert rc == 0, lib.glam_last_error()
        assert_close(out[0], sum(t[0].double().t() @ t[1].double() for t in short), 3e-6, "dw_ih of three two-row sets")
        assert_close(out[3], sum(t[2].double().sum(0) for t in short), 3e-6, "db_hh of three two-row sets")
    else:                                       # k_wgrad: a wave's row range must fit one set
        assert rc == ops._lib.GLAM_E_UNSUPPORTED
    assert lib.glam_wgrad_gemm_pair_split_seg(4, arr2(0), M, M, arr2(1), C, C, 0, p(out[0]), p(out[1]), arr2(2), M, M, arr2(3), C, C, 0, p(out[2]),
                                              p(out[3]), 2, p(ws), ws.numel(), None, None, None, None, ops._lib.stream()) == ops._lib.GLAM_E_INVALID


@pytest.mark.parametrize("N,C", [(20400, 60), (777, 44), (16, 24)])
def test_gru_forward_keeps_celu_x_for_the_backward(device, N, C):
    """glam_gru_ws_fwd_xc: the same outputs as glam_gru_ws_fwd bit for bit plus x_celu = celu(x) (src_1gp/layer.py:261) as the launch applies
    it; glam_gru_bwd_ws with celu_in = 2 on x_celu gives the gradients of celu_in = 1 on x (d_gi, d_gh, d_h, d_identity bit for bit: they do
    not involve celu'; d_x within rounding: celu'(x) = celu(x) + 1 instead of exp(x) on the negative side)."""
    lib, p = ops._lib.load(), ops._lib.ptr
    st = ops._lib.stream
    torch.manual_seed(N + C)
    M = 3 * C
    r = lambda *s: torch.randn(*s, device=device)
    x, h, idn, w_ih, w_hh, b_ih, b_hh = r(N, C), r(N, C), r(N, C), r(M, C) * 0.3, r(M, C) * 0.3, r(M), r(M)
    ia, ib = (torch.empty(lib.glam_ts_gemm_image_bytes(C, M) // 4, device=device) for _ in range(2))
    ta, tb = (torch.empty(lib.glam_ts_gemm_image_bytes(M, C) // 4, device=device) for _ in range(2))
    for w, i, t in ((w_ih, ia, ta), (w_hh, ib, tb)):
        assert lib.glam_ts_gemm_make_image(p(w), C, 1, C, M, p(i), st()) == 0
        assert lib.glam_ts_gemm_make_image(p(w), C, 0, M, C, p(t), st()) == 0
    nan = lambda *s: torch.full(s, float("nan"), device=device)
    a = [nan(N, M), nan(N, M), nan(N, C), nan(N, C)]
    b = [nan(N, M), nan(N, M), nan(N, C), nan(N, C)]
    xc = nan(N, C)
    assert lib.glam_gru_ws_fwd(p(x), p(h), p(idn), p(ia), p(ib), p(b_ih), p(b_hh), N, C, 1, 1, 0.0, p(a[0]), p(a[1]), p(a[2]), p(a[3]), st()) == 0
    assert lib.glam_gru_ws_fwd_xc(p(x), p(h), p(idn), p(ia), p(ib), p(b_ih), p(b_hh), N, C, 1, 1, 0.0, p(b[0]), p(b[1]), p(b[2]), p(b[3]), p(xc),
                                  st()) == 0, lib.glam_last_error()
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    assert_close(xc, torch.nn.functional.celu(x.double()), 2e-7, "x_celu")
    assert lib.glam_gru_ws_fwd_xc(p(x), p(h), p(idn), p(ia), p(ib), p(b_ih), p(b_hh), N, C, 0, 1, 0.0, p(b[0]), p(b[1]), p(b[2]), p(b[3]), p(xc),
                                  st()) == ops._lib.GLAM_E_INVALID                       # x_celu without the folded CELU
    d_out, d_hs = r(N, C), r(N, C)
    res = []
    for flag, xin in ((1, x), (2, xc)):
        o = [nan(N, M), nan(N, M), nan(N, C), nan(N, C), nan(N, C)]
        rc = lib.glam_gru_bwd_ws(p(a[0]), p(a[1]), p(h), p(a[3]), p(d_out), p(d_hs), p(xin), p(ta), p(tb), N, C, flag, 1, 0.0, 0, p(o[0]), p(o[1]),
                                 p(o[2]), p(o[3]), p(o[4]), st())
        assert rc == 0, lib.glam_last_error()
        res.append(o)
    for k in (0, 1, 2, 4):
        assert torch.equal(res[0][k], res[1][k])
    assert_close(res[1][3], res[0][3].double(), 3e-7, "d_x from celu(x)")


@pytest.mark.parametrize("N,C", [(20400, 60), (777, 44), (16, 24), (5000, 64), (0, 32)])
def test_gru_step_on_presplit_gate_matrices_equals_the_plain_images(device, N, C):
    """glam_gru_ws_make_pre + the *_pre entry points (the gate matrices of torch.nn.GRUCell, src_1gp/layer.py:250, as operand fragments
    split into their three bf16 terms once per weight update): every output of the forward and the backward equals, bit for bit, what
    the same kernels compute from the k_ts_gemm images (they split the same values in every block's prologue) — with and without the
    RReLU / Dropout draws, and with either image left out of the make call."""
    lib, p = ops._lib.load(), ops._lib.ptr
    st = ops._lib.stream
    torch.manual_seed(N + C)
    M = 3 * C
    r = lambda *s: torch.randn(*s, device=device)
    x, h, idn, w_ih, w_hh, b_ih, b_hh = r(N, C), r(N, C), r(N, C), r(M, C) * 0.3, r(M, C) * 0.3, r(M), r(M)
    ia, ib = (torch.empty(lib.glam_ts_gemm_image_bytes(C, M) // 4, device=device) for _ in range(2))
    ta, tb = (torch.empty(lib.glam_ts_gemm_image_bytes(M, C) // 4, device=device) for _ in range(2))
    for w, i, t in ((w_ih, ia, ta), (w_hh, ib, tb)):
        assert lib.glam_ts_gemm_make_image(p(w), C, 1, C, M, p(i), st()) == 0
        assert lib.glam_ts_gemm_make_image(p(w), C, 0, M, C, p(t), st()) == 0
    nb = lib.glam_gru_ws_pre_bytes()
    assert nb == 48 * 3 * 1024
    pre = torch.full((2, nb), 0xAB, dtype=torch.uint8, device=device)
    assert lib.glam_gru_ws_make_pre(p(w_ih), p(w_hh), C, p(pre[0]), p(pre[1]), st()) == 0, lib.glam_last_error()
    one = torch.full((2, nb), 0xAB, dtype=torch.uint8, device=device)
    assert lib.glam_gru_ws_make_pre(p(w_ih), p(w_hh), C, p(one[0]), None, st()) == 0
    assert lib.glam_gru_ws_make_pre(p(w_ih), p(w_hh), C, None, p(one[1]), st()) == 0
    assert torch.equal(one, pre)
    assert lib.glam_gru_ws_make_pre(p(w_ih), p(w_hh), 66, p(pre[0]), p(pre[1]), st()) == ops._lib.GLAM_E_UNSUPPORTED
    nan = lambda *s: torch.full(s, float("nan"), device=device)
    # forward, eval mode
    a = [nan(N, M), nan(N, M), nan(N, C), nan(N, C), nan(N, C)]
    b = [nan(N, M), nan(N, M), nan(N, C), nan(N, C), nan(N, C)]
    assert lib.glam_gru_ws_fwd_xc(p(x), p(h), p(idn), p(ia), p(ib), p(b_ih), p(b_hh), N, C, 1, 1, 0.0, p(a[0]), p(a[1]), p(a[2]), p(a[3]), p(a[4]),
                                  st()) == 0, lib.glam_last_error()
    assert lib.glam_gru_ws_fwd_pre(p(x), p(h), p(idn), p(pre[0]), p(b_ih), p(b_hh), N, C, 1, 1, 0.0, p(b[0]), p(b[1]), p(b[2]), p(b[3]), p(b[4]),
                                   st()) == 0, lib.glam_last_error()
    for u, v in zip(a, b):
        assert torch.equal(u, v) or (N == 0)
        assert N == 0 or not torch.isnan(v).any()
    if N > 0:
        assert lib.glam_gru_ws_fwd_pre(p(x), p(h), p(idn), None, p(b_ih), p(b_hh), N, C, 1, 1, 0.0, p(b[0]), p(b[1]), p(b[2]), p(b[3]), p(b[4]),
                                       st()) == ops._lib.GLAM_E_INVALID
    # backward, eval mode (x = celu(x) as the forward kept it)
    d_out, d_hs = r(N, C), r(N, C)
    res = []
    for use_pre in (False, True):
        o = [nan(N, M), nan(N, M), nan(N, C), nan(N, C), nan(N, C)]
        if use_pre:
            rc = lib.glam_gru_bwd_ws_pre(p(a[0]), p(a[1]), p(h), p(a[3]), p(d_out), p(d_hs), p(a[4]), p(pre[1]), N, C, 2, 1, 0.0, 0, p(o[0]),
                                         p(o[1]), p(o[2]), p(o[3]), p(o[4]), st())
        else:
            rc = lib.glam_gru_bwd_ws(p(a[0]), p(a[1]), p(h), p(a[3]), p(d_out), p(d_hs), p(a[4]), p(ta), p(tb), N, C, 2, 1, 0.0, 0, p(o[0]), p(o[1]),
                                     p(o[2]), p(o[3]), p(o[4]), st())
        assert rc == 0, lib.glam_last_error()
        res.append(o)
    for u, v in zip(*res):
        assert N == 0 or (torch.equal(u, v) and not torch.isnan(v).any())
    if N == 0:
        return
    # training mode: RReLU slopes and the next Dropout's mask drawn inside the launch (the same draws from the same state)
    lo, hi, dp = 0.125, 1.0 / 3.0, 0.15
    outs, effs = [], []
    for use_pre in (False, True):
        state = torch.tensor([1234] + [0] * (ops.RNG_STATE_WORDS - 1), dtype=torch.int64, device=device)
        eff = torch.zeros(2, dtype=torch.int64, device=device)
        o = [nan(N, M), nan(N, M), nan(N, C), nan(N, C), nan(N, C), nan(N, C)]
        if use_pre:
            rc = lib.glam_gru_ws_rng_fwd_pre(p(x), p(h), p(idn), p(pre[0]), p(b_ih), p(b_hh), N, C, 1, 4, 0.0, lo, hi, dp, p(state), p(eff), p(o[0]),
                                             p(o[1]), p(o[2]), p(o[3]), p(o[4]), p(o[5]), st())
        else:
            rc = lib.glam_gru_ws_rng_fwd_xc(p(x), p(h), p(idn), p(ia), p(ib), p(b_ih), p(b_hh), N, C, 1, 4, 0.0, lo, hi, dp, p(state), p(eff), p(o[0]),
                                            p(o[1]), p(o[2]), p(o[3]), p(o[4]), p(o[5]), st())
        assert rc == 0, lib.glam_last_error()
        outs.append(o); effs.append(eff)
    for u, v in zip(*outs):
        assert torch.equal(u, v) and not torch.isnan(v).any()
    assert torch.equal(effs[0], effs[1])
    d_drop = r(N, C)
    res = []
    for use_pre in (False, True):
        o = [nan(N, M), nan(N, M), nan(N, C), nan(N, C), nan(N, C)]
        f = outs[0]
        if use_pre:
            rc = lib.glam_gru_bwd_ws_rng_pre(p(f[0]), p(f[1]), p(h), p(f[3]), p(d_out), p(d_drop), p(d_hs), p(f[5]), p(pre[1]), N, C, 2, 4, 0.0, lo, hi,
                                             dp, p(effs[0]), 0, p(o[0]), p(o[1]), p(o[2]), p(o[3]), p(o[4]), st())
        else:
            rc = lib.glam_gru_bwd_ws_rng(p(f[0]), p(f[1]), p(h), p(f[3]), p(d_out), p(d_drop), p(d_hs), p(f[5]), p(ta), p(tb), N, C, 2, 4, 0.0, lo, hi,
                                         dp, p(effs[0]), 0, p(o[0]), p(o[1]), p(o[2]), p(o[3]), p(o[4]), st())
        assert rc == 0, lib.glam_last_error()
        res.append(o)
    for u, v in zip(*res):
        assert torch.equal(u, v) and not torch.isnan(v).any()


@pytest.mark.parametrize("block", ["_TripletMessage", "_NNConv"])
@pytest.mark.parametrize("steps", [1, 2, 3, 4])
def test_gru_weight_gradients_of_all_applications_in_one_launch(device, steps, block, monkeypatch, wgrad_route):
    """MessageBlock applied message_steps times: the GRU's weight gradients, the TripletMessage's parameter gradients
    (glam_triplet_layer_bwd_data_ell + glam_triplet_layer_param_grads_sets) and NNConv's relation product's (glam_wgrad_gemm_sets) as
    ONE product each over the parked operand sets of all applications (four applications: a group of three, then
    one added in place) against one product per application."""
    torch.manual_seed(steps)
    monkeypatch.setattr(ops, "USE_TORCH_EXT", False)       # the Python node: what a captured step runs (the eager C++ node parks nothing)
    b = synth_batch(160, seed=3).to(device)        # ~3 200 atoms: above the batching threshold
    net = model.Architecture(mol_block=block, message_steps=steps).to(device).eval()
    grads = {}
    for flag in (True, False):
        monkeypatch.setattr(ops, "GRU_WGRAD_BATCH", flag)
        net.zero_grad()
        out = net(b)
        out.square().sum().backward()
        grads[flag] = ({n: p.grad.clone() for n, p in net.named_parameters()}, out.detach().clone())
    assert torch.equal(grads[True][1], grads[False][1])
    for n in grads[True][0]:
        a, c = grads[True][0][n], grads[False][0][n]
        if "mol_conv" in n and steps > 1:        # GRU, TripletMessage / NNConv parameters: one product over all applications
            assert_close(a, c, 3e-6, n)          # (another summation order; ONE application: the same launches, bit for bit)
        else:
            assert torch.equal(a, c), n
    if steps > 1:     # (identical bits everywhere would mean the parked route was not taken)
        assert any(not torch.equal(grads[True][0][n], grads[False][0][n]) for n in grads[True][0] if "mol_conv.conv" in n)


@pytest.mark.parametrize("alpha,edge_dim,steps,heads", [(2, 4, 3, 3), (2, 4, 2, 3), (3, 4, 4, 3), (1, 4, 3, 3), (4, 8, 3, 3), (4, 8, 2, 3),
                                                         (2, 7, 3, 3), (4, 4, 3, 1), (4, 4, 2, 2), (2, 4, 3, 4), (4, 8, 3, 2), (1, 5, 2, 4)])
def test_triplet_parameter_gradients_over_parked_sets_on_the_general_kernels(device, alpha, edge_dim, steps, heads, monkeypatch):
    """The same for the widths and edge features the warp-specialised kernels do not take (hidden width 15 / 30 / 45: general aggregate
    kernels; continuous edge features of width 7 / 8): two and three operand sets, k_param_grads<SETS> reading 512 partial rows per
    set, against one product per application."""
    torch.manual_seed(10 * alpha + steps)
    monkeypatch.setattr(ops, "USE_TORCH_EXT", False)       # the Python node: what a captured step runs
    b = synth_batch(160, seed=5).to(device)
    if edge_dim != 4:
        b.edge_attr = torch.randn(b.edge_attr.shape[0], edge_dim, device=device)
    net = model.Architecture(mol_block="_TripletMessage", hid_dim_alpha=alpha, mol_edge_in_dim=edge_dim, message_steps=steps,
                             pre_act="ReLU", graph_act="ReLU", flat_act="ReLU", graph_do="_None()", end_do="_None()")
    if heads != 3:       # (the model's block has three heads, layer.py:239; the kernels take one to four)
        net.mol_conv.conv.conv = layer.TripletMessage(15 * alpha, edge_dim, heads=heads)
    net = net.to(device)
    grads = {}
    for flag in (True, False):
        monkeypatch.setattr(ops, "GRU_WGRAD_BATCH", flag)
        net.zero_grad()
        out = net(b)
        out.square().sum().backward()
        grads[flag] = ({n: p.grad.clone() for n, p in net.named_parameters()}, out.detach().clone())
    assert torch.equal(grads[True][1], grads[False][1])
    for n in grads[True][0]:
        assert_close(grads[True][0][n], grads[False][0][n], 3e-6, n)
    # (the one product sums in another order: identical bits everywhere would mean the parked route was not taken)
    assert any(not torch.equal(grads[True][0][n], grads[False][0][n]) for n in grads[True][0] if "mol_conv.conv" in n)


@pytest.mark.parametrize("block", ["_NNConv", "_TripletMessage"])
def test_pair_norm_and_the_dropout_behind_it_in_one_launch(device, monkeypatch, block):
    """run.py's default block (norm = _PairNorm, Dropout(0.2); src_1gp/layer.py:255-256) in training mode: the Dropout comes out of the
    norm's launch (glam_graph_norm_drop_fwd / _bwd) — the same Philox words for the same elements as the stand-alone Dropout launch, so
    the whole training step is bit-identical to the two-launch form; three launches per direction and step less."""
    from glam_amd._lib import kernel_timer
    torch.manual_seed(21)
    b = synth_batch(64, seed=9).to(device)
    net = model.Architecture(mol_block=block, graph_norm="_PairNorm", message_steps=3).to(device).train()
    monkeypatch.setattr(ops, "USE_TORCH_EXT", False)
    res = {}
    for flag in (True, False):
        monkeypatch.setattr(ops, "NORM_DROP", flag)
        ops.manual_seed(1234)
        net.zero_grad()
        with kernel_timer() as kt:
            out = net(b)
            out.square().sum().backward()
        names = [r[0] for r in kt.records()]
        res[flag] = (out.detach().clone(), [p.grad.clone() for p in net.parameters()],
                     sum("k_bias_res_act_fwd" in n for n in names), sum("k_bias_res_act_bwd" in n for n in names))
    assert torch.equal(res[True][0], res[False][0])
    for a, c in zip(res[True][1], res[False][1]):
        assert torch.equal(a, c)
    assert res[False][2] - res[True][2] == 3 and res[False][3] - res[True][3] == 3, (res[True][2:], res[False][2:])
    # C ABI: shapes outside the fused kernels are refused, p must lie in (0, 1)
    lib, p = ops._lib.load(), ops._lib.ptr
    x = torch.randn(100, 60, device=device)
    sp = ops.segment_ptr(torch.arange(100, device=device) // 10)
    y = torch.empty_like(x)
    eff = torch.empty(2, dtype=torch.int64, device=device)
    st = ops.rng_state(device)
    assert lib.glam_graph_norm_drop_fwd(p(x), p(sp.ptr), 100, 10, 60, 0, 1.0, 1e-5, 0.0, p(st), p(eff), None, p(y), ops._lib.stream()) == ops._lib.GLAM_E_INVALID
    assert lib.glam_graph_norm_drop_fwd(p(x), p(sp.ptr), 100, 1, 60, 0, 1.0, 1e-5, 0.2, p(st), p(eff), None, p(y), ops._lib.stream()) == ops._lib.GLAM_E_UNSUPPORTED
    assert lib.glam_graph_norm_drop_supported(100, 10, 60) == 1 and lib.glam_graph_norm_drop_supported(100, 10, 62) == 0
    # the plain output beside the dropped one: y_drop is y * {0, 1 / (1 - p)}
    yp = torch.empty_like(x)
    assert lib.glam_graph_norm_drop_fwd(p(x), p(sp.ptr), 100, 10, 60, 0, 1.0, 1e-5, 0.25, p(st), p(eff), p(yp), p(y), ops._lib.stream()) == 0
    ref = ops.pair_norm(x, sp)
    assert torch.equal(yp, ref)
    keep = y != 0
    assert torch.equal(y[keep], (ref * (1.0 / 0.75))[keep]) and 0.6 < keep.float().mean().item() < 0.9


@pytest.mark.parametrize("steps", [3, 4])
def test_wide_layer_parameter_gradients_of_all_applications_in_one_launch(device, steps, monkeypatch):
    """hid_dim_alpha = 6 (C = 90 -> 92: the wide TripletMessage path and the wide GRU's gate linears): the N-deep weight gradients of all
    applications as one product each over the parked operand sets (glam_wgrad_gemm_sets2, glam_wgrad_gemm_linear_sets; four
    applications: three sets + one added in place) against one product per application; forward and data gradients bit for bit."""
    torch.manual_seed(steps + 40)
    b = synth_batch(160, seed=6).to(device)
    net = model.Architecture(hid_dim_alpha=6, mol_block="_TripletMessage", message_steps=steps).to(device).eval()
    res = {}
    for flag in (True, False):
        monkeypatch.setattr(ops, "GRU_WGRAD_BATCH", flag)
        net.zero_grad()
        out = net(b)
        out.square().sum().backward()
        res[flag] = ({n: p.grad.clone() for n, p in net.named_parameters()}, out.detach().clone())
    assert torch.equal(res[True][1], res[False][1])
    for n in res[True][0]:
        a, c = res[True][0][n], res[False][0][n]
        if "mol_conv" in n:
            assert_close(a, c, 3e-6, n)
        else:
            assert torch.equal(a, c), n



@pytest.mark.parametrize("N,C,ident,celu,train", [(1, 64, True, True, False), (1000, 64, False, False, False), (20400, 60, True, True, False),
                                                  (20400, 60, True, True, True), (17, 24, True, False, True), (0, 60, True, True, False)])
def test_gru_ws_keeps_the_gates_instead_of_both_pre_activations(device, N, C, ident, celu, train):
    """gh = NULL / d_gh = NULL on the warp-specialised GRU step (torch.nn.GRU's gate equations, src_1gp/layer.py:261-266): the forward keeps
    [r | z | n | gh_n] (4C floats per row) instead of gi and gh (6C), the backward reads them and writes ONE gate-gradient matrix
    [d_pr | d_pz | d_pn | d_pn r] instead of d_gi and d_gh.  Same h_new, out, d_x, d_h, d_identity and gate gradients as the two-matrix form,
    bit for bit — they are the values the backward recomputed — in the plain and the RReLU / Dropout form."""
    raw, p, st = ops._lib.load(), ops.ptr, ops.stream
    g = torch.Generator().manual_seed(N + C + 7)
    r = lambda *s: torch.randn(*s, generator=g).to(device)
    M = 3 * C
    x, h, idn, w_ih, w_hh, b_ih, b_hh = r(N, C), r(N, C), r(N, C), r(M, C) * 0.3, r(M, C) * 0.3, r(M), r(M)
    d_out, d_hs, d_drop = r(N, C), r(N, C), r(N, C)
    nb = raw.glam_gru_ws_pre_bytes()
    pre = torch.empty(2, nb, dtype=torch.uint8, device=device)
    assert raw.glam_gru_ws_make_pre(p(w_ih), p(w_hh), C, p(pre[0]), p(pre[1]), st()) == 0
    f = lambda *s: torch.full(s, float("nan"), device=device)
    lo, hi, dp = 0.125, 1.0 / 3, 0.2
    act = 4 if train else 1
    res = []
    for gates in (False, True):
        gi, gh = (f(N, 4 * C), None) if gates else (f(N, M), f(N, M))
        hn, out, drop, xc = f(N, C), f(N, C), f(N, C), f(N, C)
        state = torch.tensor([77] + [0] * (ops.RNG_STATE_WORDS - 1), dtype=torch.int64, device=device)
        eff = torch.zeros(2, dtype=torch.int64, device=device)
        idp = p(idn) if ident else None
        if train:
            rc = raw.glam_gru_ws_rng_fwd_pre(p(x), p(h), idp, p(pre[0]), p(b_ih), p(b_hh), N, C, int(celu), act, 0.0, lo, hi, dp, p(state), p(eff),
                                             p(gi), p(gh), p(hn), p(out), p(drop), p(xc) if celu else None, st())
        else:
            rc = raw.glam_gru_ws_fwd_pre(p(x), p(h), idp, p(pre[0]), p(b_ih), p(b_hh), N, C, int(celu), act, 0.0, p(gi), p(gh), p(hn), p(out),
                                         p(xc) if celu else None, st())
        assert rc == 0, raw.glam_last_error()
        dgi, dgh = (f(N, 4 * C), None) if gates else (f(N, M), f(N, M))
        did, dx, dh = f(N, C), f(N, C), f(N, C)
        xin, flag = (xc, 2) if celu else (x, 0)
        if train:
            rc = raw.glam_gru_bwd_ws_rng_pre(p(gi), p(gh), p(h), p(out), p(d_out), p(d_drop), p(d_hs), p(xin), p(pre[1]), N, C, flag, act, 0.0, lo, hi,
                                             dp, p(eff), 0, p(dgi), p(dgh), p(did) if ident else None, p(dx), p(dh), st())
        else:
            rc = raw.glam_gru_bwd_ws_pre(p(gi), p(gh), p(h), p(out), p(d_out), p(d_hs), p(xin), p(pre[1]), N, C, flag, act, 0.0, 0, p(dgi), p(dgh),
                                         p(did) if ident else None, p(dx), p(dh), st())
        assert rc == 0, raw.glam_last_error()
        res.append((gi, gh, hn, out, drop, dgi, dgh, did, dx, dh))
    (gi, gh, hn, out, drop, dgi, dgh, did, dx, dh), (G, _, hn2, out2, drop2, D, _, did2, dx2, dh2) = res
    assert torch.equal(hn, hn2) and torch.equal(out, out2) and torch.equal(dx, dx2) and torch.equal(dh, dh2)
    assert not train or torch.equal(drop, drop2)
    assert not ident or torch.equal(did, did2)
    # the gate gradients: d_gi = D[:, :3C], d_gh = [D[:, :2C] | D[:, 3C:]]
    assert torch.equal(D[:, :M], dgi) and torch.equal(torch.cat([D[:, :2 * C], D[:, M:]], 1), dgh)
    # the gates: what the gate equations give on the two pre-activation matrices (fp64), and gh_n itself
    rr, zz = torch.sigmoid(gi[:, :C].double() + gh[:, :C].double()), torch.sigmoid(gi[:, C:2 * C].double() + gh[:, C:2 * C].double())
    nn_ = torch.tanh(gi[:, 2 * C:].double() + rr * gh[:, 2 * C:].double())
    assert_close(G[:, :C], rr, 2e-6, "r")
    assert_close(G[:, C:2 * C], zz, 2e-6, "z")
    assert_close(G[:, 2 * C:M], nn_, 2e-6, "n")
    assert torch.equal(G[:, M:], gh[:, 2 * C:])
    if N:
        # one of the pair without the other is an error, not a guess
        assert raw.glam_gru_bwd_ws_pre(p(G), None, p(h), p(out), p(d_out), p(d_hs), p(x), p(pre[1]), N, C, 0, 1, 0.0, 0, p(D), p(dgh), None, p(dx), p(dh),
                                       st()) == ops._lib.GLAM_E_INVALID
        assert raw.glam_gru_bwd_ws_pre(p(gi), p(gh), p(h), p(out), p(d_out), p(d_hs), p(x), p(pre[1]), N, C, 0, 1, 0.0, 0, p(dgi), None, None, p(dx), p(dh),
                                       st()) == ops._lib.GLAM_E_INVALID


@pytest.mark.parametrize("N,C,nseg,celu,add", [(20400, 60, 3, False, True), (20400, 60, 1, True, False), (40000, 60, 2, False, False), (700, 48, 3, False, True),
                                               (33, 24, 1, False, False), (0, 60, 1, False, False)])
def test_gru_weight_gradients_from_the_one_gate_gradient_matrix(device, N, C, nseg, celu, add):
    """glam_wgrad_gemm_gru_gates_seg: d_W_ih, d_b_ih, d_W_hh, d_b_hh of torch.nn.GRU (src_1gp/layer.py:247) from D = [d_pr | d_pz | d_pn | d_pn r]
    — glam_wgrad_gemm_pair_split_seg on the expanded d_gi = D[:, :3C], d_gh = [D[:, :2C] | D[:, 3C:]] bit for bit (same launch, same order of
    additions), and the fp64 products to rounding."""
    import ctypes
    raw, p, st = ops._lib.load(), ops.ptr, ops.stream
    g = torch.Generator().manual_seed(N + C + nseg)
    r = lambda *s: torch.randn(*s, generator=g).to(device)
    M = 3 * C
    D, X, H = [r(N, 4 * C) for _ in range(nseg)], [r(N, C) for _ in range(nseg)], [r(N, C) for _ in range(nseg)]
    dgi = [d[:, :M].contiguous() for d in D]
    dgh = [torch.cat([d[:, :2 * C], d[:, M:]], 1).contiguous() for d in D]
    adds = [r(M, C), r(M), r(M, C), r(M)] if add else [None] * 4
    arr = lambda ts: (ctypes.c_void_p * nseg)(*[t.data_ptr() for t in ts])
    ws = torch.empty(raw.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=device)
    f = lambda *s: torch.full(s, float("nan"), device=device)
    got = [f(M, C), f(M), f(M, C), f(M)]
    rc = raw.glam_wgrad_gemm_gru_gates_seg(nseg, arr(D), C, arr(X), C, int(celu), arr(H), C, p(got[0]), p(got[1]), p(got[2]), p(got[3]), N, p(ws),
                                           ws.numel(), *[p(t) for t in adds], st())
    assert rc == 0, raw.glam_last_error()
    if N == 0:
        assert all(not t.any() for t in got)
        return
    want = [f(M, C), f(M), f(M, C), f(M)]
    rc = raw.glam_wgrad_gemm_pair_split_seg(nseg, arr(dgi), M, M, arr(X), C, C, int(celu), p(want[0]), p(want[1]), arr(dgh), M, M, arr(H), C, C, 0,
                                            p(want[2]), p(want[3]), N, p(ws), ws.numel(), *[p(t) for t in adds], st())
    if rc == 0:       # (sets shorter than a wave's row range: the two-matrix form refuses, the caller runs them one by one)
        for a, b in zip(got, want):
            assert torch.equal(a, b)
    else:
        assert nseg > 1 and N < 1024
    xs = [torch.nn.functional.celu(t.double()) if celu else t.double() for t in X]
    ref = [sum(a.double().t() @ b for a, b in zip(dgi, xs)), sum(a.double().sum(0) for a in dgi),
           sum(a.double().t() @ b.double() for a, b in zip(dgh, H)), sum(a.double().sum(0) for a in dgh)]
    for a, b, t, what in zip(got, ref, adds, ("d_W_ih", "d_b_ih", "d_W_hh", "d_b_hh")):
        assert_close(a, b + (t.double() if t is not None else 0), 3e-6, what)
    assert raw.glam_wgrad_gemm_gru_gates_seg(nseg, arr(D), 62, arr(X), C, 0, arr(H), C, p(got[0]), p(got[1]), p(got[2]), p(got[3]), N, p(ws), ws.numel(),
                                             None, None, None, None, st()) == ops._lib.GLAM_E_INVALID


@pytest.mark.parametrize("train", [False, True])
def test_message_block_on_kept_gates_matches_the_two_matrix_form(device, monkeypatch, train):
    """MessageBlock (src_1gp/layer.py:240-266) applied three times with shared weights (model.py:53-54): ops.GRU_GATES on / off give the same
    outputs and the same gradients of the input and of every parameter, bit for bit (the weight gradients of all three applications in one
    launch from the three gate-gradient matrices)."""
    b = synth_batch(700, seed=5).to(device)
    torch.manual_seed(3)
    kw = dict(norm="_None", dropout="Dropout(0.2)" if train else "_None()", conv="_TripletMessage", act="RReLU" if train else "ReLU", res=True)
    blk = layer.MessageBlock(60, 60, 4, **kw).to(device)
    blk.train(train)
    x0 = torch.randn(b.x.size(0), 60, device=device)
    res = []
    for on in (False, True):
        monkeypatch.setattr(ops, "GRU_GATES", on)
        ops.manual_seed(11, device)
        x = x0.clone().requires_grad_(True)
        with ops.weight_scope():
            y, hs = x, None
            for _ in range(3):
                y, hs = blk(y, b.edge_index, b.edge_attr, h=hs, batch=b.batch)
            gs = torch.autograd.grad((y * y).sum() + hs.sum(), [x] + list(blk.parameters()))
        res.append([y, hs] + list(gs))
    for i, (a, c) in enumerate(zip(*res)):
        assert torch.equal(a, c), i


@pytest.mark.parametrize("N,C,H,ident,celu,train", [(20400, 60, 3, True, True, False), (20400, 60, 3, True, True, True), (1000, 64, 2, False, False, False),
                                                    (17, 48, 3, True, False, True), (5000, 36, 4, True, True, False), (1, 60, 3, True, True, False)])
def test_gru_step_writes_the_node_product_of_the_next_application(device, N, C, H, ident, celu, train):
    """glam_gru_ws_(rng_)fwd_pre_node: the GRU step of one application of a MessageBlock (src_1gp/layer.py:261-266) also writes
    xw | a_ij = out @ [W_node | Wa] — the first product of the block's next TripletMessage (layer.py:37; model.py:53-54 applies the same block
    message_steps times) — from the layer's staged node image.  Every other output as without it, bit for bit; the product equal to
    glam_ts_gemm's on the same rows bit for bit (the dropped twin in the rng form), and to the fp64 product to rounding."""
    raw, p, st = ops._lib.load(), ops.ptr, ops.stream
    g = torch.Generator().manual_seed(N + C + H)
    r = lambda *s: torch.randn(*s, generator=g).to(device)
    M, HC = 3 * C, H * C
    x, h, idn, w_ih, w_hh, b_ih, b_hh = r(N, C), r(N, C), r(N, C), r(M, C) * 0.3, r(M, C) * 0.3, r(M), r(M)
    wn, we, att, wsc, bias = r(C, HC) * 0.2, r(4, HC) * 0.2, r(1, H, 3 * C) * 0.2, r(HC, C) * 0.2, r(C)
    staged = torch.empty(raw.glam_triplet_staged_floats(H, C, 4), device=device)
    assert raw.glam_triplet_stage_params(p(wn), p(we), p(att), p(wsc), p(bias), C, H, 4, C, 4, p(staged), st()) == 0
    nimg, nfrag = staged[raw.glam_triplet_staged_node_image(H, C, 4):], staged[raw.glam_triplet_staged_node_fragments(H, C, 4):]
    pre = torch.empty(2, raw.glam_gru_ws_pre_bytes(), dtype=torch.uint8, device=device)
    assert raw.glam_gru_ws_make_pre(p(w_ih), p(w_hh), C, p(pre[0]), p(pre[1]), st()) == 0
    f = lambda *s: torch.full(s, float("nan"), device=device)
    lo, hi, dp = 0.125, 1.0 / 3, 0.2
    res = []
    for node in (False, True):
        G, hn, out, drop, xc, xw, a_ij = f(N, 4 * C), f(N, C), f(N, C), f(N, C), f(N, C), f(N, HC), f(N, 8)
        state = torch.tensor([91] + [0] * (ops.RNG_STATE_WORDS - 1), dtype=torch.int64, device=device)
        eff = torch.zeros(2, dtype=torch.int64, device=device)
        idp, xcp = (p(idn) if ident else None), (p(xc) if celu else None)
        tail = (p(nfrag), HC, p(xw), p(a_ij), st()) if node else (st(),)
        if train:
            fn = raw.glam_gru_ws_rng_fwd_pre_node if node else raw.glam_gru_ws_rng_fwd_pre
            rc = fn(p(x), p(h), idp, p(pre[0]), p(b_ih), p(b_hh), N, C, int(celu), 4, 0.0, lo, hi, dp, p(state), p(eff), p(G), None, p(hn), p(out),
                    p(drop), xcp, *tail)
        else:
            fn = raw.glam_gru_ws_fwd_pre_node if node else raw.glam_gru_ws_fwd_pre
            rc = fn(p(x), p(h), idp, p(pre[0]), p(b_ih), p(b_hh), N, C, int(celu), 1, 0.0, p(G), None, p(hn), p(out), xcp, *tail)
        assert rc == 0, raw.glam_last_error()
        res.append((G, hn, out, drop, xc, xw, a_ij))
    for u, v in zip(res[0][:5], res[1][:5]):
        assert torch.equal(u, v) or (torch.isnan(u).all() and torch.isnan(v).all())
    rows = res[1][3] if train else res[1][2]
    xw, a_ij = res[1][5], res[1][6]
    want_xw, want_a = f(N, HC), f(N, 8)
    assert raw.glam_ts_gemm(p(rows), C, C, None, 0, 0, p(nimg), None, p(want_xw), HC, HC, p(want_a), 8, 8, N, st()) == 0, raw.glam_last_error()
    if ops._lib.route_enabled("x3"):       # (GLAM_X3=0 puts glam_ts_gemm on the fp32 matrix instructions: equal to rounding then)
        assert torch.equal(xw, want_xw) and torch.equal(a_ij, want_a)
    assert not torch.isnan(xw).any() and not torch.isnan(a_ij).any()
    assert_close(xw, want_xw, 2e-6, "xw vs glam_ts_gemm")
    assert_close(a_ij, want_a, 2e-6, "a_ij vs glam_ts_gemm")
    wa = torch.zeros(C, 8, dtype=torch.float64)
    for hh in range(H):
        wa[:, hh] = wn.double().cpu()[:, hh * C:(hh + 1) * C] @ att.double().cpu()[0, hh, :C]
        wa[:, 4 + hh] = wn.double().cpu()[:, hh * C:(hh + 1) * C] @ att.double().cpu()[0, hh, 2 * C:]
    assert_close(xw, rows.double().cpu() @ wn.double().cpu(), 2e-6, "xw")
    assert_close(a_ij, rows.double().cpu() @ wa, 2e-6, "a_ij")
    # all of (image, xw, a_ij) or none; widths outside the image table
    assert raw.glam_gru_ws_fwd_pre_node(p(x), p(h), None, p(pre[0]), p(b_ih), p(b_hh), N, C, 0, 1, 0.0, p(G), None, p(hn), p(out), None, p(nfrag), HC, None,
                                        p(a_ij), st()) == ops._lib.GLAM_E_INVALID
    assert raw.glam_gru_ws_fwd_pre_node(p(x), p(h), None, p(pre[0]), p(b_ih), p(b_hh), N, C, 0, 1, 0.0, p(G), None, p(hn), p(out), None, p(nfrag), 40, p(xw),
                                        p(a_ij), st()) == ops._lib.GLAM_E_UNSUPPORTED


@pytest.mark.parametrize("train,B", [(False, 700), (True, 700), (False, 3)])
def test_model_step_with_the_node_product_inside_the_gru_step(device, monkeypatch, train, B):
    """Architecture (src_1gp/model.py:36-62) forward + backward with ops.NODE_IN_GRU on / off: the same outputs and the same gradients bit for
    bit — the product a TripletMessage finds ready is the one its own launch would write — and three launches less per step (all three node
    GEMMs: the first application's comes out of the input embedding's launch)."""
    b = synth_batch(B, seed=8).to(device)
    torch.manual_seed(5)
    kw = dict(mol_block="_TripletMessage", hid_dim_alpha=4, e_dim=256, out_dim=1, message_steps=3, mol_readout="GlobalPool5")
    if not train:
        kw.update(pre_act="ReLU", graph_act="ReLU", flat_act="ReLU", graph_do="_None()", end_do="_None()")
    from glam_amd import graphs
    from glam_amd._lib import kernel_timer
    net = model.Architecture(**kw).to(device)
    net.train(train)
    monkeypatch.setattr(graphs, "GRAPHED_CALL", False)      # (eager launches: the timer sees them)
    monkeypatch.setattr(ops, "USE_TORCH_EXT", False)        # (the Python nodes: the route of a captured step)
    res, counts = [], []
    for on in (False, True):
        monkeypatch.setattr(ops, "NODE_IN_GRU", on)
        ops.manual_seed(11, device)
        net.zero_grad(set_to_none=True)
        with kernel_timer(capacity=256) as kt:
            y = net(b)
            y.square().sum().backward()
        names = [k for k, _, _ in kt.records()]
        counts.append((sum("k_ts_gemm" in k for k in names), sum("+node" in k for k in names)))
        res.append([y.detach().clone()] + [q.grad.clone() for q in net.parameters()])
    for i, (u, v) in enumerate(zip(*res)):
        assert torch.equal(u, v), i
    if B >= 16 and ops._lib.route_enabled("x3") and ops._lib.load().glam_triplet_layer_ws_supported(3, 60, 4, 1) == 1 and ops.GRU_PRE:
        # (the warp-specialised GRU step and layer: GLAM_X3=0 / GLAM_WS=0 / GLAM_GRU_PRE=0 switch the route off; a handful of molecules: one
        #  tile per block all the same)
        # (the embedding's launch writes the first application's product, the GRU steps of the first two the second's and third's)
        want = 3 if (not train or ops.RRELU_IN_GEMM) else 2      # (training mode: the embedding's product carries it only with its RReLU in the epilogue)
        assert counts[0][1] == 0 and counts[1][1] == want and counts[0][0] - counts[1][0] == want, counts


@pytest.mark.parametrize("N,K,M,p", [(20400, 16, 60, 0.2), (20400, 16, 60, 0.0), (17, 32, 64, 0.5), (1, 16, 28, 0.2), (0, 16, 60, 0.2)])
def test_rrelu_and_the_dropped_twin_in_the_embedding_products_epilogue(device, N, K, M, p):
    """glam_ts_gemm_rrelu (the LinearBlock of src_1gp/layer.py:223-237 with the reference's default activation, model.py:31, in training
    mode): the same bits as glam_ts_gemm followed by glam_bias_res_act_rng_fwd from the same stream position — output, dropped twin, the
    recorded (seed, offset) pair and the position the stream is left at — and glam_bias_res_act_rng_bwd on its output is its backward."""
    raw, ptr, st = ops._lib.load(), ops.ptr, ops.stream
    if not ops._lib.route_enabled("x3"):      # (GLAM_X3=0: no k_tall_x3, no such epilogue — the two-launch form is what runs)
        assert raw.glam_ts_gemm_rrelu_supported(K, M) == 0
        return
    g = torch.Generator().manual_seed(N + K + M)
    r = lambda *s: torch.randn(*s, generator=g).to(device)
    x, w, b = r(N, K), r(M, K) * 0.5, r(M)
    img = torch.empty(raw.glam_ts_gemm_image_bytes(K, M) // 4, device=device)
    assert raw.glam_ts_gemm_make_image(ptr(w), K, 1, K, M, ptr(img), st()) == 0
    f = lambda *s: torch.full(s, float("nan"), device=device)
    lo, hi = 0.125, 1.0 / 3
    mk = lambda: (torch.tensor([4242] + [0] * (ops.RNG_STATE_WORDS - 1), dtype=torch.int64, device=device), torch.zeros(2, dtype=torch.int64, device=device))
    # two launches
    s0, e0 = mk()
    y0, o0, d0 = f(N, M), f(N, M), f(N, M)
    assert raw.glam_ts_gemm(ptr(x), K, K, None, 0, 0, ptr(img), ptr(b), ptr(y0), M, M, None, 0, 0, N, st()) == 0
    assert raw.glam_bias_res_act_rng_fwd(ptr(y0), None, None, N, M, 4, 0.0, lo, hi, p, ptr(s0), ptr(e0), ptr(o0), ptr(d0) if p > 0 else None, st()) == 0
    # one
    s1, e1 = mk()
    o1, d1 = f(N, M), f(N, M)
    assert raw.glam_ts_gemm_rrelu(ptr(x), K, K, ptr(img), ptr(b), M, N, lo, hi, p, ptr(s1), ptr(e1), ptr(o1), ptr(d1) if p > 0 else None, st()) == 0, raw.glam_last_error()
    if N == 0:
        return
    assert torch.equal(o0, o1) and not torch.isnan(o1).any()
    assert p == 0 or (torch.equal(d0, d1) and 0.0 < (d1 == 0).float().mean().item() < 1.0 or N * M < 64)
    assert torch.equal(e0, e1) and torch.equal(s0[:2], s1[:2]) and int(s1[1]) == 1        # the next launch draws from the next position
    assert (o1 < 0).any() or N * M < 8                                                     # (negative inputs keep a slope: not a ReLU)
    # widths outside the epilogue's table
    assert raw.glam_ts_gemm_rrelu_supported(16, 60) == 1 and raw.glam_ts_gemm_rrelu_supported(128, 60) == 0
    assert raw.glam_ts_gemm_rrelu(ptr(x), K, K, ptr(img), ptr(b), M, N, 0.0, hi, p, ptr(s1), ptr(e1), ptr(o1), None, st()) == ops._lib.GLAM_E_INVALID


@pytest.mark.parametrize("next_p", [0.0, 0.2])
def test_linear_block_with_training_mode_rrelu_in_one_launch(device, monkeypatch, next_p):
    """LinearBlock(15 -> 60, act RReLU).train() — mol_lin0 of the reference's default model (model.py:40, :49) — with ops.RRELU_IN_GEMM on /
    off: the same output, dropped twin and gradients bit for bit, one launch less."""
    from glam_amd._lib import kernel_timer
    torch.manual_seed(2)
    blk = layer.LinearBlock(15, 60, act="RReLU").to(device).train()
    x = torch.randn(5000, 15, device=device)
    res, counts = [], []
    for on in (False, True):
        monkeypatch.setattr(ops, "RRELU_IN_GEMM", on)
        ops.manual_seed(7, device)
        blk.zero_grad(set_to_none=True)
        with ops.weight_scope(), kernel_timer(capacity=64) as kt:
            y = blk(x, next_dropout=next_p)
            twin = ops.take_dropped(y, next_p) if next_p else None
            ((y * y).sum() + (twin.sum() if twin is not None else 0)).backward()
        counts.append(len(kt.records()))
        res.append([y.detach(), None if twin is None else twin.detach(), blk.linear.weight.grad.clone(), blk.linear.bias.grad.clone()])
    for u, v in zip(*res):
        assert (u is None and v is None) or torch.equal(u, v)
    assert (next_p == 0) == (res[0][1] is None)
    if ops._lib.route_enabled("x3"):
        assert counts[0] - counts[1] == 1, counts


@pytest.mark.parametrize("N,K,M,H,act,p", [(20400, 16, 60, 3, 1, 0.0), (20400, 16, 60, 3, 4, 0.2), (20400, 16, 60, 3, 4, 0.0), (5000, 16, 60, 3, 0, 0.0),
                                           (17, 32, 64, 2, 4, 0.5), (1, 16, 48, 3, 1, 0.0), (70, 16, 36, 4, 4, 0.2), (0, 16, 60, 3, 1, 0.0)])
def test_embedding_launch_writes_the_node_product_of_the_first_application(device, N, K, M, H, act, p):
    """glam_ts_gemm_act_node: the input embedding (LinearBlock, src_1gp/model.py:40, :49) with none / ReLU / training-mode RReLU (+ dropped
    twin) AND the node product of the TripletMessage behind it (layer.py:37) in one launch — output, twin, stream position as
    glam_ts_gemm(_relu / _rrelu) writes them, xw | a_ij as glam_ts_gemm on those rows (the twin when there is one), all bit for bit."""
    raw, ptr, st = ops._lib.load(), ops.ptr, ops.stream
    if not ops._lib.route_enabled("x3"):
        assert raw.glam_ts_gemm_rrelu_supported(K, M) == 0
        return
    g = torch.Generator().manual_seed(N + K + M + act)
    r = lambda *s: torch.randn(*s, generator=g).to(device)
    HC = H * M
    x, w, b = r(N, K), r(M, K) * 0.5, r(M)
    wn, we, att, wsc, bias = r(M, HC) * 0.2, r(4, HC) * 0.2, r(1, H, 3 * M) * 0.2, r(HC, M) * 0.2, r(M)
    staged = torch.empty(raw.glam_triplet_staged_floats(H, M, 4), device=device)
    assert raw.glam_triplet_stage_params(ptr(wn), ptr(we), ptr(att), ptr(wsc), ptr(bias), M, H, 4, M, 4, ptr(staged), st()) == 0
    nimg, nfrag = staged[raw.glam_triplet_staged_node_image(H, M, 4):], staged[raw.glam_triplet_staged_node_fragments(H, M, 4):]
    img = torch.empty(raw.glam_ts_gemm_image_bytes(K, M) // 4, device=device)
    assert raw.glam_ts_gemm_make_image(ptr(w), K, 1, K, M, ptr(img), st()) == 0
    f = lambda *s: torch.full(s, float("nan"), device=device)
    lo, hi = 0.125, 1.0 / 3
    mk = lambda: (torch.tensor([99] + [0] * (ops.RNG_STATE_WORDS - 1), dtype=torch.int64, device=device), torch.zeros(2, dtype=torch.int64, device=device))
    s0, e0 = mk()
    o0, d0 = f(N, M), f(N, M)
    if act == 4:
        assert raw.glam_ts_gemm_rrelu(ptr(x), K, K, ptr(img), ptr(b), M, N, lo, hi, p, ptr(s0), ptr(e0), ptr(o0), ptr(d0) if p > 0 else None, st()) == 0
    elif act == 1:
        assert raw.glam_ts_gemm_relu(ptr(x), K, K, ptr(img), ptr(b), ptr(o0), M, M, N, st()) == 0
    else:
        assert raw.glam_ts_gemm(ptr(x), K, K, None, 0, 0, ptr(img), ptr(b), ptr(o0), M, M, None, 0, 0, N, st()) == 0
    rows = d0 if (act == 4 and p > 0) else o0
    xw0, a0 = f(N, HC), f(N, 8)
    assert raw.glam_ts_gemm(ptr(rows), M, M, None, 0, 0, ptr(nimg), None, ptr(xw0), HC, HC, ptr(a0), 8, 8, N, st()) == 0
    s1, e1 = mk()
    o1, d1, xw1, a1 = f(N, M), f(N, M), f(N, HC), f(N, 8)
    rc = raw.glam_ts_gemm_act_node(ptr(x), K, K, ptr(img), ptr(b), M, N, act, lo, hi, p, ptr(s1) if act == 4 else None, ptr(e1) if act == 4 else None,
                                   ptr(o1), ptr(d1) if (act == 4 and p > 0) else None, ptr(nfrag), HC, ptr(xw1), ptr(a1), st())
    assert rc == 0, raw.glam_last_error()
    if N == 0:
        return
    assert torch.equal(o0, o1) and not torch.isnan(o1).any()
    if act == 4:
        assert torch.equal(e0, e1) and torch.equal(s0[:2], s1[:2]) and (p == 0 or torch.equal(d0, d1))
    if ops._lib.route_enabled("x3"):
        assert torch.equal(xw0, xw1) and torch.equal(a0, a1)
    assert_close(xw1, xw0, 2e-6, "xw")
    assert_close(a1, a0, 2e-6, "a_ij")
    assert not torch.isnan(xw1).any() and not torch.isnan(a1).any()
    assert raw.glam_ts_gemm_act_node(ptr(x), K, K, ptr(img), ptr(b), M, N, 2, lo, hi, p, None, None, ptr(o1), None, ptr(nfrag), HC, ptr(xw1), ptr(a1),
                                     st()) == ops._lib.GLAM_E_UNSUPPORTED


@pytest.mark.parametrize("N,K,M,p", [(1024, 1024, 1, 0.2), (32, 1024, 1, 0.2), (300, 256, 12, 0.5), (1000, 64, 2, 0.0), (3, 1024, 16, 0.2), (0, 1024, 1, 0.2)])
def test_head_applies_the_hidden_layers_rrelu_and_its_own_dropout(device, N, K, M, p):
    """glam_linear_narrow_act_fwd / _bwd: y = Linear(Dropout(p)(RReLU(x))) in training mode with x the PRE-activation of mol_flat
    (src_1gp/model.py:43-47, :60-61; layer.py:232-236) — the bits of the three-launch pipeline (glam_bias_res_act_rng_fwd writing the
    activated matrix and its dropped twin, then glam_linear_narrow_fwd on the twin) forward and backward, from the same stream position."""
    raw, ptr, st = ops._lib.load(), ops.ptr, ops.stream
    g = torch.Generator().manual_seed(N + K + M)
    r = lambda *s: torch.randn(*s, generator=g).to(device)
    x, w, b, dy = r(N, K), r(M, K) * 0.1, r(M), r(N, M)
    f = lambda *s: torch.full(s, float("nan"), device=device)
    lo, hi = 0.125, 1.0 / 3
    mk = lambda: (torch.tensor([5] + [0] * (ops.RNG_STATE_WORDS - 1), dtype=torch.int64, device=device), torch.zeros(2, dtype=torch.int64, device=device))
    ws = torch.empty(raw.glam_linear_narrow_bwd_workspace_bytes(K, M), dtype=torch.uint8, device=device)
    # three launches forward, two backward
    s0, e0 = mk()
    act, twin, y0 = f(N, K), f(N, K), f(N, M)
    assert raw.glam_bias_res_act_rng_fwd(ptr(x), None, None, N, K, 4, 0.0, lo, hi, p, ptr(s0), ptr(e0), ptr(act), ptr(twin) if p > 0 else None, st()) == 0
    rows = twin if p > 0 else act
    assert raw.glam_linear_narrow_fwd(ptr(rows), ptr(w), ptr(b), N, K, M, ptr(y0), st()) == 0
    d_rows, dw0, db0, dx0 = f(N, K), f(M, K), f(M), f(N, K)
    assert raw.glam_linear_narrow_bwd(ptr(rows), ptr(w), ptr(dy), N, K, M, ptr(d_rows), ptr(dw0), ptr(db0), ptr(ws), ws.numel(), st()) == 0
    if N:
        assert raw.glam_bias_res_act_rng_bwd(ptr(act), ptr(d_rows) if p == 0 else None, ptr(d_rows) if p > 0 else None, N, K, 4, 0.0, lo, hi, p, ptr(e0),
                                             ptr(dx0), st()) == 0
    # one each way
    s1, e1 = mk()
    y1, dx1, dw1, db1 = f(N, M), f(N, K), f(M, K), f(M)
    assert raw.glam_linear_narrow_act_fwd(ptr(x), ptr(w), ptr(b), N, K, M, lo, hi, p, ptr(s1), ptr(e1), ptr(y1), st()) == 0, raw.glam_last_error()
    assert raw.glam_linear_narrow_act_bwd(ptr(x), ptr(w), ptr(dy), N, K, M, lo, hi, p, ptr(e1), ptr(dx1), ptr(dw1), ptr(db1), ptr(ws), ws.numel(), st()) == 0
    if N == 0:
        assert not dw1.any() and not db1.any()
        return
    assert torch.equal(y0, y1) and torch.equal(e0, e1) and int(s1[1]) == 1
    assert torch.equal(dw0, dw1) and torch.equal(db0, db1) and torch.equal(dx0, dx1) and not torch.isnan(dx1).any()
    assert raw.glam_linear_narrow_act_fwd(ptr(x), ptr(w), ptr(b), N, K, M, 0.0, hi, p, ptr(s1), ptr(e1), ptr(y1), st()) == ops._lib.GLAM_E_INVALID


def test_default_model_head_without_an_activated_matrix(device, monkeypatch):
    """Architecture() defaults in train() (model.py:24-33): ops.HEAD_ACT_FUSED on / off — same outputs and gradients bit for bit, four
    launches less per step (mol_flat's RReLU + twin, its backward, folded into the head's two launches)."""
    from glam_amd import graphs
    from glam_amd._lib import kernel_timer
    b = synth_batch(200, seed=4).to(device)
    torch.manual_seed(9)
    net = model.Architecture(mol_block="_TripletMessage", e_dim=256).to(device).train()
    monkeypatch.setattr(graphs, "GRAPHED_CALL", False)
    monkeypatch.setattr(ops, "USE_TORCH_EXT", False)
    res, counts = [], []
    for on in (False, True):
        monkeypatch.setattr(ops, "HEAD_ACT_FUSED", on)
        ops.manual_seed(3, device)
        net.zero_grad(set_to_none=True)
        with kernel_timer(capacity=256) as kt:
            y = net(b)
            y.square().sum().backward()
        names = [k for k, _, _ in kt.records()]
        counts.append((sum("k_bias_res_act" in k for k in names), sum("rrelu+dropout" in k for k in names)))
        res.append([y.detach().clone()] + [q.grad.clone() for q in net.parameters()])
    for i, (u, v) in enumerate(zip(*res)):
        assert torch.equal(u, v), i
    # (the stand-alone RReLU launches of mol_flat, forward and backward, against the head's two launches doing their work)
    assert counts[0][0] - counts[1][0] == 2 and counts[0][1] == 0 and counts[1][1] == 2, counts


@pytest.mark.parametrize("arch", ["ddi", "dti"])
def test_two_input_models_with_triplet_towers_and_node_products(device, monkeypatch, arch):
    """ArchitectureDDI / ArchitectureDTI (src_2gi_ddi/model.py, src_2gi_dti_scr/model.py) with `_TripletMessage` ligand towers: the GRU
    steps of both towers write the next application's node product (the per-pair fusion hands the rows back untouched); ops.NODE_IN_GRU
    on / off give the same outputs and gradients bit for bit."""
    from glam_amd import graphs
    monkeypatch.setattr(graphs, "GRAPHED_CALL", False)
    monkeypatch.setattr(ops, "USE_TORCH_EXT", False)
    torch.manual_seed(1)
    kw = dict(mol_block="_TripletMessage", message_steps=3, e_dim=128, pre_act="ReLU", graph_act="ReLU", flat_act="ReLU", end_act="ReLU",
              graph_do="_None()", end_do="_None()")
    if arch == "ddi":
        net = model.ArchitectureDDI(**kw).to(device)
        a, b = synth_batch(300, seed=1).to(device), synth_batch(300, seed=2).to(device)
    else:
        net = model.ArchitectureDTI(pro_block="_GCNConv", **kw).to(device)
        a, b = synth_batch(6, seed=1).to(device), synth_protein_batch(6, seed=2, n_min=40, n_max=90).to(device)
    res = []
    for on in (False, True):
        monkeypatch.setattr(ops, "NODE_IN_GRU", on)
        net.zero_grad(set_to_none=True)
        y = net(a, b)
        y.square().sum().backward()
        res.append([y.detach().clone()] + [q.grad.clone() for q in net.parameters() if q.grad is not None])
    assert len(res[0]) == len(res[1]) > 5
    for i, (u, v) in enumerate(zip(*res)):
        assert torch.equal(u, v), i
