"""TEST INFRASTRUCTURE — golden-vector generator (runs ONLY in the build container).

Imports the reference's own ``src_1gp/layer.py`` / ``model.py`` (and the two-graph
``src_2gi_dti_scr`` / ``src_2gi_ddi`` ``layer.py`` / ``model.py``) from ``/root/reference`` over ``oracle/pyg_standin``, runs them
on seeded synthetic inputs, and writes inputs + parameters + outputs + autograd gradients
to ``tests/golden/*.npz``.  While doing so it checks the restatement in
``oracle/glam_oracle.py`` against the reference (outputs and gradients) — that is what
"pins" the oracle.  Nothing from /root/reference is copied: the fixtures are data.

    python oracle/gen_goldens.py            # regenerate + verify
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "pyg_standin"))

from glam_amd.data import synth_batch, synth_protein_batch, Batch, Data  # noqa: E402
import oracle.glam_oracle as O  # noqa: E402

TOL = 2e-6


def load_reference(subdir):
    """Import ``layer`` and ``model`` of one reference variant under private names."""
    d = os.path.join(REF, subdir)
    sys.path.insert(0, d)
    try:
        for name in ("layer", "model"):
            sys.modules.pop(name, None)
        spec = importlib.util.spec_from_file_location("layer", os.path.join(d, "layer.py"))
        layer = importlib.util.module_from_spec(spec)
        sys.modules["layer"] = layer
        spec.loader.exec_module(layer)
        spec = importlib.util.spec_from_file_location("model", os.path.join(d, "model.py"))
        model = importlib.util.module_from_spec(spec)
        sys.modules["model"] = model
        spec.loader.exec_module(model)
    finally:
        sys.path.remove(d)
        sys.modules.pop("layer", None)
        sys.modules.pop("model", None)
    return layer, model


def seed(s):
    torch.manual_seed(s)
    np.random.seed(s)


def grads_of(out, cot, tensors):
    loss = (out * cot).sum()
    gs = torch.autograd.grad(loss, tensors, allow_unused=True)
    return [torch.zeros_like(t) if g is None else g for g, t in zip(gs, tensors)]


ONLY = os.environ.get("GLAM_GOLDEN_ONLY")      # write only the fixtures whose name contains this (everything is still computed and checked)


def save(name, meta, inputs, params, out, cot, grads):
    if ONLY and ONLY not in name:
        return
    arrs = {"meta": np.array(json.dumps(meta))}
    for k, v in inputs.items():
        arrs["in." + k] = v.detach().numpy()
    for k, v in params.items():
        arrs["param." + k] = v.detach().numpy()
    arrs["out"] = out.detach().numpy()
    if cot is not None:
        arrs["cot"] = cot.numpy()
    for k, v in grads.items():
        arrs["grad." + k] = v.detach().numpy()
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.1f} KiB)")


GTOL = 5e-6      # gradients of single layers / blocks (scaled); whole-model gradients: 1e-5 (deeper chains of fp32 roundings)


def check(tag, a, b, tol=TOL):
    """Oracle vs reference: max |d| scaled by max(1, |ref|) must stay within `tol`; returns the SCALED error."""
    err = (a - b).abs().max().item() if a.numel() else 0.0
    scale = max(1.0, b.abs().max().item() if b.numel() else 1.0)
    assert err <= tol * scale, f"{tag}: oracle vs reference max|d|={err:.3e} (scale {scale:.2f}, tol {tol:.0e})"
    return err / scale


# ----------------------------------------------------------------------------------
# graph builders for edge cases
# ----------------------------------------------------------------------------------
def hidden_batch(B, C, sd, De=4):
    """ESOL-shaped topology with N(0,1) hidden features of width C."""
    b = synth_batch(B, seed=sd)
    g = torch.Generator().manual_seed(1000 + sd)
    b.x = torch.randn(b.x.size(0), C, generator=g)
    if De != 4:
        b.edge_attr = torch.rand(b.edge_attr.size(0), De, generator=g)
    return b


def edge_case_batch(C, De):
    """One batch holding: a single-atom graph (isolated node), a 2-node graph (sort-pool
    pad), duplicate edges, a star with in-degree 70, unsorted edge order, continuous edge_attr."""
    g = torch.Generator().manual_seed(77)
    graphs = []
    graphs.append(Data(torch.randn(1, C, generator=g), torch.zeros(2, 0, dtype=torch.long), torch.zeros(0, De)))
    graphs.append(Data(torch.randn(2, C, generator=g), torch.tensor([[0, 1], [1, 0]]), torch.rand(2, De, generator=g)))
    ei = torch.tensor([[0, 1, 1, 2, 0, 1, 3], [1, 0, 2, 1, 1, 0, 1]])  # (0->1),(1->0) duplicated; node 3 has no in-edge
    graphs.append(Data(torch.randn(4, C, generator=g), ei, torch.rand(7, De, generator=g)))
    n = 71
    src = torch.arange(1, n)
    ei = torch.cat([torch.stack([src, torch.zeros_like(src)]), torch.stack([torch.zeros_like(src), src])], 1)
    perm = torch.randperm(ei.size(1), generator=g)
    graphs.append(Data(torch.randn(n, C, generator=g), ei[:, perm], torch.rand(ei.size(1), De, generator=g)))
    b = Batch.from_data_list(graphs)
    return b


# ----------------------------------------------------------------------------------
def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(4)
    layer, model = load_reference("src_1gp")
    layer2, model2 = load_reference("src_2gi_dti_scr")
    worst = 0.0

    # ---- TripletMessage: channel sweep on ESOL-shaped batches + edge cases ----------------
    print("TripletMessage")
    cases = [("c%d" % C, hidden_batch(6 if C == 60 else 3, C, C), C, 4) for C in (15, 30, 45, 60, 90)]
    cases.append(("de8", hidden_batch(3, 60, 5, De=8), 60, 8))
    cases.append(("edge", edge_case_batch(60, 4), 60, 4))
    cases.append(("edge_c15_de8", edge_case_batch(15, 8), 15, 8))
    for tag, b, C, De in cases:
        seed(11 + C + De)
        conv = layer.TripletMessage(C, De)
        with torch.no_grad():
            conv.bias.normal_(0, 0.1)   # reset_parameters zeros it; make the add visible
        x = b.x.clone().requires_grad_(True)
        ea = b.edge_attr.clone().requires_grad_(True)
        out = conv(x, b.edge_index, ea)
        cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(5))
        names = [n for n, _ in conv.named_parameters()]
        ps = [p for _, p in conv.named_parameters()]
        gs = grads_of(out, cot, [x, ea] + ps)
        # oracle check (forward + backward)
        xo = b.x.clone().requires_grad_(True)
        eo = b.edge_attr.clone().requires_grad_(True)
        po = [p.detach().clone().requires_grad_(True) for p in ps]
        oo = O.triplet_message(xo, b.edge_index, eo, *po)
        go = grads_of(oo, cot, [xo, eo] + po)
        worst = max(worst, check(f"triplet/{tag}/out", oo, out))
        for n_, a, r in zip(["x", "edge_attr"] + names, go, gs):
            worst = max(worst, check(f"triplet/{tag}/grad.{n_}", a, r, GTOL))
        # aggregate-level intermediates for op-level kernel tests
        xw = torch.matmul(b.x, conv.weight_node).detach()
        ew = torch.matmul(b.edge_attr, conv.weight_edge).detach()
        aggr = O.triplet_aggregate(xw, b.edge_index, ew, conv.weight_triplet_att.detach(), 3)
        save(f"triplet_{tag}", {"C": C, "De": De, "heads": 3, "slope": 0.2, "kind": "TripletMessage"},
             {"x": b.x, "edge_index": b.edge_index, "edge_attr": b.edge_attr, "batch": b.batch},
             dict(zip(names, ps)), out, cot,
             dict(zip(["x", "edge_attr"] + names, gs)) | {"__aggr": aggr})

    # ---- TripletMessageLight ----------------------------------------------------------------
    print("TripletMessageLight")
    for tag, b, C, De in [("c60", hidden_batch(6, 60, 21), 60, 4), ("c45", hidden_batch(3, 45, 22), 45, 4),
                          ("edge", edge_case_batch(60, 4), 60, 4), ("edge_c30_de8", edge_case_batch(30, 8), 30, 8)]:
        seed(31 + C)
        conv = layer.TripletMessageLight(C, De)
        with torch.no_grad():
            conv.bias.normal_(0, 0.1)
        x = b.x.clone().requires_grad_(True)
        ea = b.edge_attr.clone().requires_grad_(True)
        out = conv(x, b.edge_index, ea)
        cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(6))
        names = [n for n, _ in conv.named_parameters()]
        ps = [p for _, p in conv.named_parameters()]
        gs = grads_of(out, cot, [x, ea] + ps)
        xo = b.x.clone().requires_grad_(True)
        eo = b.edge_attr.clone().requires_grad_(True)
        po = [p.detach().clone().requires_grad_(True) for p in ps]
        oo = O.triplet_message_light(xo, b.edge_index, eo, *po)
        go = grads_of(oo, cot, [xo, eo] + po)
        worst = max(worst, check(f"light/{tag}/out", oo, out))
        for n_, a, r in zip(["x", "edge_attr"] + names, go, gs):
            worst = max(worst, check(f"light/{tag}/grad.{n_}", a, r, GTOL))
        save(f"light_{tag}", {"C": C, "De": De, "slope": 0.2, "kind": "TripletMessageLight"},
             {"x": b.x, "edge_index": b.edge_index, "edge_attr": b.edge_attr, "batch": b.batch},
             dict(zip(names, ps)), out, cot, dict(zip(["x", "edge_attr"] + names, gs)))

    # ---- readouts -----------------------------------------------------------------------------
    print("readouts")
    for tag, b in [("esol", hidden_batch(6, 60, 41)), ("edge", edge_case_batch(60, 4)),
                   ("c15", hidden_batch(5, 15, 42))]:
        B = int(b.batch.max()) + 1
        x = b.x.clone().requires_grad_(True)
        out = layer.GlobalPool5()(x, b.batch)
        cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(7))
        (gx,) = grads_of(out, cot, [x])
        xo = b.x.clone().requires_grad_(True)
        oo = O.global_pool5(xo, b.batch, B)
        (go,) = grads_of(oo, cot, [xo])
        worst = max(worst, check(f"pool5/{tag}/out", oo, out), check(f"pool5/{tag}/gx", go, gx))
        mx = layer.global_max_pool(b.x, b.batch)
        worst = max(worst, check(f"maxpool/{tag}", O.global_max_pool(b.x, b.batch, B), mx))
        save(f"pool5_{tag}", {"kind": "GlobalPool5", "B": B}, {"x": b.x, "batch": b.batch}, {}, out, cot,
             {"x": gx, "__max": mx})

    for tag, b in [("esol", hidden_batch(6, 60, 43)), ("edge", edge_case_batch(60, 4))]:
        B = int(b.batch.max()) + 1
        seed(44)
        pool = layer.GlobalLAPool(60)
        x = b.x.clone().requires_grad_(True)
        out = pool(x, b.batch)
        cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(8))
        names = [n for n, _ in pool.named_parameters()]
        ps = [p for _, p in pool.named_parameters()]
        gs = grads_of(out, cot, [x] + ps)
        sd = pool.state_dict()
        oo = O.global_attention(b.x, b.batch, B, sd["pool.gate_nn.weight"], sd["pool.gate_nn.bias"],
                                sd["pool.nn.weight"], sd["pool.nn.bias"])
        worst = max(worst, check(f"lapool/{tag}", oo, out))
        save(f"lapool_{tag}", {"kind": "GlobalLAPool", "B": B}, {"x": b.x, "batch": b.batch},
             dict(zip(names, ps)), out, cot, dict(zip(["x"] + names, gs)))

    b = hidden_batch(6, 60, 45)
    B = 6
    seed(46)
    from torch_geometric.nn import Set2Set
    s2s = Set2Set(60, processing_steps=3)
    x = b.x.clone().requires_grad_(True)
    out = s2s(x, b.batch)
    cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(9))
    names = [n for n, _ in s2s.named_parameters()]
    ps = [p for _, p in s2s.named_parameters()]
    gs = grads_of(out, cot, [x] + ps)
    worst = max(worst, check("set2set", O.set2set(b.x, b.batch, B, s2s.lstm), out))
    save("set2set_esol", {"kind": "Set2Set", "B": B, "steps": 3}, {"x": b.x, "batch": b.batch},
         dict(zip(names, ps)), out, cot, dict(zip(["x"] + names, gs)))

    # ---- norms ----------------------------------------------------------------------------------
    print("norms")
    b = edge_case_batch(60, 4)
    B = int(b.batch.max()) + 1
    x = b.x.clone().requires_grad_(True)
    outs, gxs = {}, {}
    cot = torch.randn(b.x.shape, generator=torch.Generator().manual_seed(10))
    for nm, mod, orc in [
        ("pair", layer._PairNorm(60), lambda t: O.pair_norm(t, b.batch, B)),
        ("pair_nobatch", None, lambda t: O.pair_norm(t)),
        ("layer", layer._LayerNorm(60), None),
        ("gsize", layer._GraphSizeNorm(60), lambda t: O.graph_size_norm(t, None)),
    ]:
        if nm == "pair_nobatch":
            out = layer._PairNorm(60)(x, None)
        else:
            out = mod(x, b.batch)
        if nm == "layer":
            orc = lambda t, m=mod: O.graph_layer_norm(t, m.norm.weight.detach(), m.norm.bias.detach(), b.batch, B)  # noqa
        (gx,) = grads_of(out, cot, [x])
        worst = max(worst, check(f"norm/{nm}", orc(b.x), out))
        outs[nm], gxs[nm] = out.detach(), gx
    save("norms_edge", {"kind": "norms", "B": B}, {"x": b.x, "batch": b.batch}, {}, outs["pair"], cot,
         {f"__out_{k}": v for k, v in outs.items()} | {f"__gx_{k}": v for k, v in gxs.items()})

    # ---- MessageBlock (+GRU, residual) --------------------------------------------------------
    print("MessageBlock")
    for tag, conv, norm, act in [("triplet_relu", "_TripletMessage", "_None", "ReLU"),
                                 ("triplet_pair_rrelu", "_TripletMessage", "_PairNorm", "RReLU"),
                                 ("light_celu", "_TripletMessageLight", "_None", "CELU"),
                                 ("nnconv_relu", "_NNConv", "_None", "ReLU"),
                                 ("gcn_relu", "_GCNConv", "_None", "ReLU"),
                                 ("gat_leaky", "_GATConv", "_LayerNorm", "LeakyReLU")]:
        b = hidden_batch(5, 60, 51)
        B = 5
        seed(52)
        blk = layer.MessageBlock(60, 60, 4, norm=norm, dropout="_None()", conv=conv, act=act, res=True).eval()
        x = b.x.clone().requires_grad_(True)
        x1, h1 = blk(x, b.edge_index, b.edge_attr, h=None, batch=b.batch)
        x2, h2 = blk(x1, b.edge_index, b.edge_attr, h=h1, batch=b.batch)   # second step, shared weights
        out = x2
        cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(12))
        names = [n for n, _ in blk.named_parameters()]
        ps = [p for _, p in blk.named_parameters()]
        gs = grads_of(out, cot, [x] + ps)
        sd = {k: v.detach() for k, v in blk.state_dict().items()}
        o1, g1 = O.message_block(sd, "", b.x, b.edge_index, b.edge_attr, None, b.batch, B, conv, norm, act)
        o2, g2 = O.message_block(sd, "", o1, b.edge_index, b.edge_attr, g1, b.batch, B, conv, norm, act)
        worst = max(worst, check(f"block/{tag}", o2, out, 5e-6), check(f"block/{tag}/h", g2, h2.squeeze(0), 5e-6))
        save(f"block_{tag}", {"kind": "MessageBlock", "conv": conv, "norm": norm, "act": act, "B": B, "steps": 2},
             {"x": b.x, "edge_index": b.edge_index, "edge_attr": b.edge_attr, "batch": b.batch},
             dict(zip(names, ps)), out, cot, dict(zip(["x"] + names, gs)) | {"__h": h2.squeeze(0), "__x1": x1})

    # ---- Architecture (eval mode) + 3 Adam steps ----------------------------------------------
    print("Architecture")
    for tag, kw in [("triplet_pool5", dict(mol_block="_TripletMessage", mol_readout="GlobalPool5")),
                    ("light_lapool", dict(mol_block="_TripletMessageLight", mol_readout="GlobalLAPool")),
                    # the reference's own defaults (run.py:21,25,28): NNConv blocks, PairNorm, GlobalPool5
                    ("nnconv_pairnorm_pool5", dict(mol_block="_NNConv", mol_readout="GlobalPool5", graph_norm="_PairNorm"))]:
        b = synth_batch(8, seed=61)
        B = 8
        seed(62)
        net = model.Architecture(e_dim=64, out_dim=2, message_steps=3, **kw).eval()
        out = net(b)
        cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(13))
        names = [n for n, _ in net.named_parameters()]
        ps = [p for _, p in net.named_parameters()]
        gs = grads_of(out, cot, ps)
        sd = {k: v.detach() for k, v in net.state_dict().items()}
        oo = O.architecture(sd, b, B, 3, kw["mol_block"], kw["mol_readout"], graph_norm=kw.get("graph_norm", "_None"))
        worst = max(worst, check(f"arch/{tag}", oo, out, 1e-5))
        # three optimiser steps exactly as TrainerMolRegression.train_iterations
        # (src_1gp/trainer.py:286-298) but in eval mode (no dropout / RReLU noise)
        net2 = model.Architecture(e_dim=64, out_dim=1, message_steps=3, **kw).eval()
        net2.load_state_dict({k: (v if "lin_out1" not in k else v[:1]) for k, v in sd.items()})
        opt = torch.optim.Adam(net2.parameters(), lr=1e-3)
        y = b.y.view(-1)
        trace = []
        for _ in range(3):
            opt.zero_grad()
            loss = torch.nn.MSELoss()(net2(b).view(-1), y)
            loss.backward()
            gn = torch.sqrt(sum((p.grad ** 2).sum() for p in net2.parameters()))
            opt.step()
            trace.append([loss.item(), gn.item()])
        save(f"arch_{tag}", {"kind": "Architecture", "B": B, "e_dim": 64, "out_dim": 2, "message_steps": 3, **kw},
             {"x": b.x, "edge_index": b.edge_index, "edge_attr": b.edge_attr, "batch": b.batch, "y": b.y},
             dict(zip(names, ps)), out, cot,
             dict(zip(names, gs)) | {"__train_trace": torch.tensor(trace), "__post_out": net2(b).detach()})

    # ---- two-graph fusion (src_2gi_dti_scr/layer.py:270-283) ----------------------------------
    print("dot_and_global_pool2/5")
    mb = hidden_batch(4, 60, 71)
    pb = synth_protein_batch(4, seed=72, n_min=30, n_max=60)
    pb.x = torch.randn(pb.x.size(0), 60, generator=torch.Generator().manual_seed(73))
    mo = mb.x.clone().requires_grad_(True)
    po = pb.x.clone().requires_grad_(True)
    out2 = layer2.dot_and_global_pool2(mo, po, mb.batch, pb.batch)
    cot = torch.randn(out2.shape, generator=torch.Generator().manual_seed(14))
    gm, gp = grads_of(out2, cot, [mo, po])
    m5 = mb.x.clone().requires_grad_(True)
    p5 = pb.x.clone().requires_grad_(True)
    out5 = layer.dot_and_global_pool5(m5, p5, mb.batch, pb.batch)          # src_1gp/layer.py:270-283, executed as is
    cot5 = torch.randn(out5.shape, generator=torch.Generator().manual_seed(15))
    g5m, g5p = grads_of(out5, cot5, [m5, p5])
    mo5, po5 = mb.x.clone().requires_grad_(True), pb.x.clone().requires_grad_(True)
    oo5 = O.dot_and_global_pool(mo5, po5, mb.batch, pb.batch, 4, 5)
    go5m, go5p = grads_of(oo5, cot5, [mo5, po5])
    worst = max(worst, check("dot2", O.dot_and_global_pool(mb.x, pb.x, mb.batch, pb.batch, 4, 2), out2),
                check("dot5", oo5, out5, GTOL), check("dot5/g_mol", go5m, g5m, GTOL), check("dot5/g_pro", go5p, g5p, GTOL))
    save("dotpool_pairs", {"kind": "dot_and_global_pool", "B": 4},
         {"mol_x": mb.x, "pro_x": pb.x, "mol_batch": mb.batch, "pro_batch": pb.batch}, {}, out2, cot,
         {"mol_x": gm, "pro_x": gp, "__out5": out5.detach(), "__cot5": cot5, "__g5_mol": g5m, "__g5_pro": g5p})

    # ---- two-tower model (src_2gi_dti_scr/model.py:14-68), the reference's default blocks ----------
    print("Architecture (two towers: ligand + protein)")
    seed(91)
    kw = dict(mol_block="_NNConv", pro_block="_GCNConv", graph_norm="_None", pre_act="ReLU", graph_act="ReLU", flat_act="ReLU",
              end_act="ReLU")
    net = model2.Architecture(e_dim=64, message_steps=2, graph_do="_None()", end_do="_None()", **kw).eval()
    mb = synth_batch(4, seed=81)
    pb = synth_protein_batch(4, seed=82, n_min=30, n_max=90)
    names = [n for n, _ in net.named_parameters()]
    out = net(mb, pb)
    cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(15))
    gs = grads_of(out, cot, [p for _, p in net.named_parameters()])
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    o_ref = O.architecture_dti(sd, mb, pb, 4, message_steps=2, **kw)
    worst = max(worst, check("dti out", o_ref, out, 5e-6))
    for n, g_o, g_r in zip(names, grads_of(o_ref, cot, [sd[n] for n in names]), gs):
        worst = max(worst, check("dti grad " + n, g_o, g_r, 1e-5))
    save("dti_nnconv_gcn", {"kind": "ArchitectureDTI", "B": 4, "e_dim": 64, "message_steps": 2, **kw},
         {"mol_x": mb.x, "mol_edge_index": mb.edge_index, "mol_edge_attr": mb.edge_attr, "mol_batch": mb.batch,
          "pro_x": pb.x, "pro_edge_index": pb.edge_index, "pro_edge_attr": pb.edge_attr, "pro_batch": pb.batch},
         {k: v for k, v in net.state_dict().items()}, out, cot, dict(zip(names, gs)))

    # ---- two-drug model (src_2gi_ddi/model.py:9-62): two ligand towers, both block types of interest ----------
    print("Architecture (two towers: ligand + ligand)")
    layer3, model3 = load_reference("src_2gi_ddi")
    # (seed 99 for the attention block: with ReLU outputs the last channel — GlobalPool5's sort key — is zero for many atoms, and about
    #  half of the seeds put tied atoms among the top rows, where the reference's answer is torch.sort's tie order, not arithmetic)
    for tag, blk, sd_, alpha in (("nnconv", "_NNConv", 93, 2), ("triplet", "_TripletMessage", 99, 4)):
        seed(sd_)
        kw = dict(mol_block=blk, graph_norm="_None", pre_act="ReLU", graph_act="ReLU", flat_act="ReLU", end_act="ReLU")
        net = model3.Architecture(e_dim=64, message_steps=2, hid_dim_alpha=alpha, graph_do="_None()", end_do="_None()", **kw).eval()
        m1 = synth_batch(4, seed=83)
        m2 = synth_batch(4, seed=84)
        names = [n for n, _ in net.named_parameters()]
        out = net(m1, m2)
        cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(16))
        gs = grads_of(out, cot, [p for _, p in net.named_parameters()])
        sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
        o_ref = O.architecture_ddi(sd, m1, m2, 4, message_steps=2, **kw)
        worst = max(worst, check("ddi out", o_ref, out, 5e-6))
        for n, g_o, g_r in zip(names, grads_of(o_ref, cot, [sd[n] for n in names]), gs):
            worst = max(worst, check("ddi grad " + n, g_o, g_r, 1e-5))
        save("ddi_" + tag, {"kind": "ArchitectureDDI", "B": 4, "e_dim": 64, "message_steps": 2, "hid_dim_alpha": alpha, **kw},
             {"mol1_x": m1.x, "mol1_edge_index": m1.edge_index, "mol1_edge_attr": m1.edge_attr, "mol1_batch": m1.batch,
              "mol2_x": m2.x, "mol2_edge_index": m2.edge_index, "mol2_edge_attr": m2.edge_attr, "mol2_batch": m2.batch},
             {k: v for k, v in net.state_dict().items()}, out, cot, dict(zip(names, gs)))

    print(f"oracle pinned against the reference: worst scaled max|d| = {worst:.3e}")


if __name__ == "__main__":
    main()
