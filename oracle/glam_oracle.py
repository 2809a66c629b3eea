"""TEST INFRASTRUCTURE — CPU oracle for the GLAM message-passing path.

A plain-torch (CPU, fp32 or fp64) *restatement* of the reference algorithm, written
"reference-shaped": materialised neighbour gathers, concatenated triplets, scatter based
segment softmax and scatter-add aggregation — the same op sequence the reference executes
through PyG, without PyG.  Each function cites the reference lines it follows
(paths relative to /root/reference).

Pinned: ``oracle/gen_goldens.py`` (run in the build container, where /root/reference is
mounted) imports the reference's own ``src_1gp/layer.py`` / ``model.py`` over the
``oracle/pyg_standin`` package, checks every function below against the reference's
outputs *and autograd gradients*, and writes the golden vectors under ``tests/golden/``;
``tests/test_oracle_golden.py`` re-checks this file against those vectors everywhere.
The reference ships no tests or golden vectors of its own, and its
third-party base classes (torch_geometric 1.7.2) are restated, not executed — that part of
the parity chain rests on the builder's reading of PyG's published semantics.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this module.  The product package ``glam_amd`` never does.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# scatter / segment primitives (torch_scatter.scatter, torch_geometric.utils.softmax)
# --------------------------------------------------------------------------------------
def scatter(src, index, dim_size, reduce="sum"):
    """``torch_scatter.scatter(src, index, dim=0, dim_size, reduce)``; empty segment -> 0."""
    shape = (dim_size,) + tuple(src.shape[1:])
    idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    if reduce in ("sum", "add"):
        return src.new_zeros(shape).scatter_add_(0, idx, src)
    if reduce == "mean":
        tot = src.new_zeros(shape).scatter_add_(0, idx, src)
        cnt = src.new_zeros(dim_size).scatter_add_(0, index, src.new_ones(index.shape)).clamp_(min=1)
        return tot / cnt.view((-1,) + (1,) * (src.dim() - 1))
    if reduce == "max":
        return src.new_zeros(shape).scatter_reduce(0, idx, src, reduce="amax", include_self=False)
    raise ValueError(reduce)


def segment_softmax(src, index, num_segments):
    """PyG ``utils.softmax`` as called at src_1gp/layer.py:51 and :95 — max-shifted exp,
    ``+1e-16`` in the denominator, every trailing column independently."""
    m = scatter(src, index, num_segments, "max").index_select(0, index)
    p = (src - m).exp()
    s = scatter(p, index, num_segments, "sum").index_select(0, index)
    return p / (s + 1e-16)


# --------------------------------------------------------------------------------------
# TripletMessage (src_1gp/layer.py:15-64)
# --------------------------------------------------------------------------------------
def triplet_aggregate(xw, edge_index, ew, att, heads, slope=0.2):
    """PyG ``propagate`` + ``TripletMessage.message`` + add-aggregate
    (src_1gp/layer.py:40-55): ``xw[N,H*C]`` transformed nodes, ``ew[E,H*C]`` transformed
    edges, ``att[1,H,3C]``.  Returns ``aggr[N,H,C]`` (before ``update``)."""
    N = xw.size(0)
    C = xw.size(1) // heads
    src, dst = edge_index[0], edge_index[1]
    x_j = xw.index_select(0, src).view(-1, heads, C)        # __lift__ with edge_index[0]
    x_i = xw.index_select(0, dst).view(-1, heads, C)        # __lift__ with edge_index[1]
    e_ij = ew.view(-1, heads, C)
    triplet = torch.cat([x_i, e_ij, x_j], dim=-1)           # layer.py:48
    alpha = (triplet * att).sum(dim=-1)                     # layer.py:49
    alpha = F.leaky_relu(alpha, slope)                      # layer.py:50
    alpha = segment_softmax(alpha, dst, N)                  # layer.py:51
    msg = alpha.view(-1, heads, 1) * e_ij * x_j             # layer.py:55
    return scatter(msg, dst, N, "sum")                      # aggr='add' (layer.py:17)


def triplet_message(x, edge_index, edge_attr, weight_node, weight_edge, weight_triplet_att,
                    weight_scale, bias, heads=3, slope=0.2):
    """``TripletMessage.forward`` + ``update`` (src_1gp/layer.py:36-61)."""
    xw = torch.matmul(x, weight_node)                       # layer.py:37
    ew = torch.matmul(edge_attr, weight_edge)               # layer.py:38
    aggr = triplet_aggregate(xw, edge_index, ew, weight_triplet_att, heads, slope)
    out = torch.matmul(aggr.reshape(x.size(0), weight_scale.size(0)), weight_scale)   # layer.py:58-59 (explicit width: N = 0 is legal)
    return out + bias                                       # layer.py:60


# --------------------------------------------------------------------------------------
# TripletMessageLight (src_1gp/layer.py:67-104)
# --------------------------------------------------------------------------------------
def triplet_light_aggregate(xw, edge_index, edge_attr, att, slope=0.2):
    """``TripletMessageLight.message`` + add-aggregate (layer.py:88-97): raw ``edge_attr``
    enters the logit only, the message is ``alpha * x_j``."""
    N = xw.size(0)
    src, dst = edge_index[0], edge_index[1]
    x_j, x_i = xw.index_select(0, src), xw.index_select(0, dst)
    triplet = torch.cat([x_i, edge_attr, x_j], dim=-1)      # layer.py:92
    alpha = (triplet * att).sum(dim=-1)                     # layer.py:93
    alpha = F.leaky_relu(alpha, slope)
    alpha = segment_softmax(alpha, dst, N)                  # layer.py:95
    return scatter(alpha.view(-1, 1) * x_j, dst, N, "sum")  # layer.py:96-97


def triplet_message_light(x, edge_index, edge_attr, weight_node, weight_triplet_att, bias, slope=0.2):
    xw = torch.matmul(x, weight_node)                       # layer.py:84
    return triplet_light_aggregate(xw, edge_index, edge_attr, weight_triplet_att, slope) + bias  # :99-101


# --------------------------------------------------------------------------------------
# NNConv with aggr='mean' (the reference's default block, src_1gp/layer.py:115-122)
# --------------------------------------------------------------------------------------
def nnconv_mean(x, edge_index, edge_attr, nn_w0, nn_b0, nn_w1, nn_b1, root, bias):
    """``NNConv(C, C, Linear(De,32)-ReLU-Linear(32,C*C), aggr='mean')``: per-edge weight
    ``W_e = nn(edge_attr).view(C, C)``, message ``x_j @ W_e``, scatter-mean, + root + bias."""
    N, C = x.shape
    src, dst = edge_index[0], edge_index[1]
    hidden = F.relu(F.linear(edge_attr, nn_w0, nn_b0))
    w_e = F.linear(hidden, nn_w1, nn_b1).view(-1, C, root.size(1))
    msg = torch.matmul(x.index_select(0, src).unsqueeze(1), w_e).squeeze(1)
    return scatter(msg, dst, N, "mean") + torch.matmul(x, root) + bias


# --------------------------------------------------------------------------------------
# GCNConv / GATConv with PyG 1.7.2 defaults (wrapped at src_1gp/layer.py:143-158)
# --------------------------------------------------------------------------------------
def _with_self_loops(edge_index, N):
    mask = edge_index[0] != edge_index[1]
    loop = torch.arange(N, dtype=edge_index.dtype).unsqueeze(0).repeat(2, 1)
    return torch.cat([edge_index[:, mask], loop], dim=1)


def gcn_conv(x, edge_index, weight, bias):
    """``GCNConv(in, out)``: ``D^-1/2 (A + I) D^-1/2 X W + b`` (add_remaining_self_loops, unit edge weights)."""
    N = x.size(0)
    ei = _with_self_loops(edge_index, N)
    row, col = ei[0], ei[1]
    deg = scatter(x.new_ones(ei.size(1)), col, N, "sum")
    dis = deg.pow(-0.5)
    dis = dis.masked_fill(dis == float("inf"), 0)
    norm = dis[row] * dis[col]
    xw = torch.matmul(x, weight)
    return scatter(norm.view(-1, 1) * xw.index_select(0, row), col, N, "sum") + bias


def gat_conv(x, edge_index, lin_weight, att_l, att_r, bias, slope=0.2):
    """``GATConv(in, out)`` with heads=1: self loops replaced, ``alpha = softmax_i(leaky(a_l[j] + a_r[i]))``."""
    N = x.size(0)
    ei = _with_self_loops(edge_index, N)
    src, dst = ei[0], ei[1]
    xl = F.linear(x, lin_weight)
    al = (xl * att_l.view(1, -1)).sum(-1)
    ar = (xl * att_r.view(1, -1)).sum(-1)
    alpha = F.leaky_relu(al[src] + ar[dst], slope)
    alpha = segment_softmax(alpha.view(-1, 1), dst, N)
    return scatter(alpha * xl.index_select(0, src), dst, N, "sum") + bias


# --------------------------------------------------------------------------------------
# Readouts (src_1gp/layer.py:197-220, model.py:41)
# --------------------------------------------------------------------------------------
def global_add_pool(x, batch, num_graphs):
    return scatter(x, batch, num_graphs, "sum")


def global_mean_pool(x, batch, num_graphs):
    return scatter(x, batch, num_graphs, "mean")


def global_max_pool(x, batch, num_graphs):
    return scatter(x, batch, num_graphs, "max")


def global_sort_pool(x, batch, num_graphs, k):
    """PyG ``global_sort_pool``: per graph, the ``k`` rows with the largest LAST channel in
    descending order (ties keep node order — stable), zero padded; ``[B, k*D]``."""
    D = x.size(1)
    out = x.new_zeros(num_graphs, k, D)
    for g in range(num_graphs):
        rows = (batch == g).nonzero().view(-1)
        if rows.numel() == 0:
            continue
        order = torch.sort(x[rows, -1], descending=True, stable=True).indices[:k]
        out[g, : order.numel()] = x[rows[order]]
    return out.view(num_graphs, k * D)


def global_pool5(x, batch, num_graphs):
    """``GlobalPool5.forward`` (layer.py:201-203): mean || add || sort-pool(k=3)."""
    return torch.cat([global_mean_pool(x, batch, num_graphs), global_add_pool(x, batch, num_graphs),
                      global_sort_pool(x, batch, num_graphs, 3)], dim=-1)


def global_attention(x, batch, num_graphs, gate_w, gate_b, nn_w, nn_b):
    """``GlobalLAPool`` = PyG ``GlobalAttention(Linear(C,1), Linear(C,2C))`` (layer.py:206-220)."""
    gate = segment_softmax(F.linear(x, gate_w, gate_b).view(-1, 1), batch, num_graphs)
    return scatter(gate * F.linear(x, nn_w, nn_b), batch, num_graphs, "sum")


def set2set(x, batch, num_graphs, lstm, steps=3):
    """PyG ``Set2Set(C, processing_steps=3)`` (model.py:41); ``lstm`` = ``torch.nn.LSTM(2C, C)``."""
    C = x.size(1)
    h = (x.new_zeros(1, num_graphs, C), x.new_zeros(1, num_graphs, C))
    q_star = x.new_zeros(num_graphs, 2 * C)
    for _ in range(steps):
        q, h = lstm(q_star.unsqueeze(0), h)
        q = q.view(num_graphs, C)
        e = (x * q[batch]).sum(dim=-1, keepdim=True)
        a = segment_softmax(e, batch, num_graphs)
        q_star = torch.cat([q, scatter(a * x, batch, num_graphs, "sum")], dim=-1)
    return q_star


# --------------------------------------------------------------------------------------
# Graph norms (src_1gp/layer.py:161-194)
# --------------------------------------------------------------------------------------
def pair_norm(x, batch=None, num_graphs=None, scale=1.0, eps=1e-5):
    if batch is None:
        x = x - x.mean(dim=0, keepdim=True)
        return scale * x / (eps + x.pow(2).sum(-1).mean()).sqrt()
    x = x - scatter(x, batch, num_graphs, "mean")[batch]
    sq = scatter(x.pow(2).sum(-1, keepdim=True), batch, num_graphs, "mean")
    return scale * x / (eps + sq[batch]).sqrt()


def graph_layer_norm(x, weight, bias, batch=None, num_graphs=None, eps=1e-5):
    if batch is None:
        x = x - x.mean()
        out = x / (x.std(unbiased=False) + eps)
    else:
        cnt = scatter(x.new_ones(x.size(0)), batch, num_graphs, "sum").clamp_(min=1) * x.size(1)
        mean = scatter(x, batch, num_graphs, "sum").sum(-1, keepdim=True) / cnt.view(-1, 1)
        x = x - mean[batch]
        var = scatter(x * x, batch, num_graphs, "sum").sum(-1, keepdim=True) / cnt.view(-1, 1)
        out = x / (var + eps).sqrt()[batch]
    return out * weight + bias


def graph_size_norm(x, batch=None, num_graphs=None):
    if batch is None:
        return x * (x.size(0) ** -0.5)
    cnt = scatter(x.new_ones(x.size(0)), batch, num_graphs, "sum")
    return x * cnt.pow(-0.5)[batch].view(-1, 1)


# --------------------------------------------------------------------------------------
# Pairwise fusion of the two-graph variants (src_2gi_dti_scr/layer.py:270-283,
# src_1gp/layer.py:270-283)
# --------------------------------------------------------------------------------------
def dot_and_global_pool(mol_out, pro_out, mol_batch, pro_batch, num_pairs, stats=2):
    out = mol_out.new_zeros(num_pairs, stats)
    for i in range(num_pairs):
        item = torch.matmul(mol_out[mol_batch == i], pro_out[pro_batch == i].T)
        if stats == 2:
            out[i] = torch.stack([item.max(), item.mean()])
        else:
            out[i] = torch.stack([item.max(), item.mean(), item.median(), item.min(), item.std()])
    return out


# --------------------------------------------------------------------------------------
# MessageBlock / Architecture (src_1gp/layer.py:240-267, model.py:23-62) for the
# deterministic configuration used in parity tests: dropout off, no RReLU noise.
# --------------------------------------------------------------------------------------
RRELU_EVAL_SLOPE = (1.0 / 8 + 1.0 / 3) / 2  # torch.nn.RReLU() in eval mode


def activation(name, x, prelu_weight=None):
    name = name.replace("()", "")
    if name == "_None":
        return x
    if name == "ReLU":
        return F.relu(x)
    if name == "CELU":
        return F.celu(x)
    if name == "LeakyReLU":
        return F.leaky_relu(x, 0.01)
    if name == "RReLU":
        return F.leaky_relu(x, RRELU_EVAL_SLOPE)
    if name == "PReLU":
        return F.prelu(x, prelu_weight)
    raise ValueError(name)


def gru_step(x, h, w_ih, w_hh, b_ih, b_hh):
    """One step of ``torch.nn.GRU(C, C)`` with seq_len 1 (layer.py:247, :262)."""
    gi = F.linear(x, w_ih, b_ih)
    gh = F.linear(h, w_hh, b_hh)
    i_r, i_z, i_n = gi.chunk(3, dim=1)
    h_r, h_z, h_n = gh.chunk(3, dim=1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return (1 - z) * n + z * h


def message_block(sd, prefix, x, edge_index, edge_attr, h, batch, num_graphs, conv="_TripletMessage",
                  norm="_None", act="ReLU", res=True):
    """``MessageBlock.forward`` (layer.py:252-267), eval mode (dropout = identity).
    ``sd`` is the reference state dict; ``prefix`` e.g. ``'mol_conv.'``."""
    identity = x
    if h is None:
        h = x                                                # layer.py:254 (pre-norm x)
    if norm == "_PairNorm":
        x = pair_norm(x, batch, num_graphs)
    elif norm == "_LayerNorm":
        x = graph_layer_norm(x, sd[prefix + "norm.norm.weight"], sd[prefix + "norm.norm.bias"], batch, num_graphs)
    elif norm == "_GraphSizeNorm":
        x = graph_size_norm(x, None)                         # _GraphSizeNorm drops batch (layer.py:193-194)
    elif norm != "_None":
        raise ValueError(norm)
    p = prefix + "conv.conv."
    if conv == "_TripletMessage":
        x = triplet_message(x, edge_index, edge_attr, sd[p + "weight_node"], sd[p + "weight_edge"],
                            sd[p + "weight_triplet_att"], sd[p + "weight_scale"], sd[p + "bias"])
    elif conv == "_TripletMessageLight":
        x = triplet_message_light(x, edge_index, edge_attr, sd[p + "weight_node"],
                                  sd[p + "weight_triplet_att"], sd[p + "bias"])
    elif conv == "_NNConv":
        x = nnconv_mean(x, edge_index, edge_attr, sd[p + "nn.0.weight"], sd[p + "nn.0.bias"],
                        sd[p + "nn.2.weight"], sd[p + "nn.2.bias"], sd[p + "root"], sd[p + "bias"])
    elif conv == "_GCNConv":
        x = gcn_conv(x, edge_index, sd[p + "weight"], sd[p + "bias"])
    elif conv == "_GATConv":
        x = gat_conv(x, edge_index, sd[p + "lin_l.weight"], sd[p + "att_l"], sd[p + "att_r"], sd[p + "bias"])
    else:
        raise ValueError(conv)
    if conv not in ("_GCNConv", "_GATConv"):                 # layer.py:248: no GRU for GCN / GAT
        g = prefix + "gru."
        x = F.celu(x)                                        # layer.py:261
        h = gru_step(x, h, sd[g + "weight_ih_l0"], sd[g + "weight_hh_l0"], sd[g + "bias_ih_l0"], sd[g + "bias_hh_l0"])
        x = h                                                # layer.py:262-263
    if res is not False:                                     # layer.py:265 tests `self.res is False`: graph_res = 0 (an int,
        x = x + identity                                     # run.py:38 / glam.py:81) still adds the residual, as in the reference
    return activation(act, x), h


def linear_block(sd, prefix, x, act):
    """``LinearBlock.forward`` (layer.py:232-237) with ``_None`` norm / dropout."""
    return activation(act, F.linear(x, sd[prefix + "linear.weight"], sd[prefix + "linear.bias"]))


def architecture(sd, data, num_graphs, message_steps=3, mol_block="_TripletMessage", mol_readout="GlobalPool5",
                 graph_norm="_None", pre_act="RReLU", graph_act="RReLU", flat_act="RReLU", graph_res=True):
    """``Architecture.forward`` (model.py:47-62), eval mode."""
    xm = linear_block(sd, "mol_lin0.", data.x, pre_act)                          # model.py:49
    hm = None
    for _ in range(message_steps):                                               # model.py:53-54
        xm, hm = message_block(sd, "mol_conv.", xm, data.edge_index, data.edge_attr, hm, data.batch,
                               num_graphs, conv=mol_block, norm=graph_norm, act=graph_act, res=graph_res)
    if mol_readout == "GlobalPool5":                                             # model.py:57
        out = global_pool5(xm, data.batch, num_graphs)
    elif mol_readout == "GlobalLAPool":
        q = "mol_readout.pool."
        out = global_attention(xm, data.batch, num_graphs, sd[q + "gate_nn.weight"], sd[q + "gate_nn.bias"],
                               sd[q + "nn.weight"], sd[q + "nn.bias"])
    else:
        raise ValueError(mol_readout)
    out = linear_block(sd, "mol_flat.", out, flat_act)                           # model.py:60
    return linear_block(sd, "lin_out1.", out, "_None")                           # model.py:61


def architecture_dti(sd, mol, pro, num_pairs, message_steps=3, mol_block="_NNConv", pro_block="_GCNConv", graph_norm="_None",
                     pre_act="RReLU", graph_act="RReLU", flat_act="RReLU", end_act="RReLU", graph_res=True):
    """Two-tower ``Architecture.forward`` (src_2gi_dti_scr/model.py:45-68), eval mode, GlobalPool5 readouts: ligand and
    protein towers stepped side by side, ``dot_and_global_pool2`` of the two node sets after every message step, readouts,
    ``lin_out0`` on ``[outm | outp | fusion]``, ``lin_out1``."""
    xm = linear_block(sd, "mol_lin0.", mol.x, pre_act)                           # :47
    xp = linear_block(sd, "pro_lin0.", pro.x, pre_act)                           # :48
    hm = hp = None
    fusion = []
    for _ in range(message_steps):                                               # :53-56
        xm, hm = message_block(sd, "mol_conv.", xm, mol.edge_index, mol.edge_attr, hm, mol.batch, num_pairs, conv=mol_block,
                               norm=graph_norm, act=graph_act, res=graph_res)
        xp, hp = message_block(sd, "pro_conv.", xp, pro.edge_index, pro.edge_attr, hp, pro.batch, num_pairs, conv=pro_block,
                               norm=graph_norm, act=graph_act, res=graph_res)
        fusion.append(dot_and_global_pool(xm, xp, mol.batch, pro.batch, num_pairs, stats=2))
    outm = linear_block(sd, "mol_flat.", global_pool5(xm, mol.batch, num_pairs), flat_act)    # :59-60
    outp = linear_block(sd, "pro_flat.", global_pool5(xp, pro.batch, num_pairs), flat_act)    # :61-62
    out = torch.cat([outm, outp, torch.cat(fusion, dim=-1)], dim=-1)             # :65
    return linear_block(sd, "lin_out1.", linear_block(sd, "lin_out0.", out, end_act), "_None")


def architecture_ddi(sd, mol1, mol2, num_pairs, message_steps=3, mol_block="_NNConv", graph_norm="_None",
                     pre_act="RReLU", graph_act="RReLU", flat_act="RReLU", end_act="RReLU", graph_res=True):
    """Two-drug ``Architecture.forward`` (src_2gi_ddi/model.py:40-62), eval mode, GlobalPool5 readouts: two LIGAND towers with their own
    parameters (``mol1_*`` / ``mol2_*``, the same block type) stepped side by side, ``dot_and_global_pool2`` of the two node sets after
    every message step, readouts, ``lin_out0`` on ``[outm1 | outm2 | fusion]``, ``lin_out1``."""
    x1 = linear_block(sd, "mol1_lin0.", mol1.x, pre_act)                          # :42
    x2 = linear_block(sd, "mol2_lin0.", mol2.x, pre_act)                          # :43
    h1 = h2 = None
    fusion = []
    for _ in range(message_steps):                                               # :48-51
        x1, h1 = message_block(sd, "mol1_conv.", x1, mol1.edge_index, mol1.edge_attr, h1, mol1.batch, num_pairs, conv=mol_block,
                               norm=graph_norm, act=graph_act, res=graph_res)
        x2, h2 = message_block(sd, "mol2_conv.", x2, mol2.edge_index, mol2.edge_attr, h2, mol2.batch, num_pairs, conv=mol_block,
                               norm=graph_norm, act=graph_act, res=graph_res)
        fusion.append(dot_and_global_pool(x1, x2, mol1.batch, mol2.batch, num_pairs, stats=2))
    o1 = linear_block(sd, "mol1_flat.", global_pool5(x1, mol1.batch, num_pairs), flat_act)    # :54-57
    o2 = linear_block(sd, "mol2_flat.", global_pool5(x2, mol2.batch, num_pairs), flat_act)
    out = torch.cat([o1, o2, torch.cat(fusion, dim=-1)], dim=-1)                 # :60
    return linear_block(sd, "lin_out1.", linear_block(sd, "lin_out0.", out, end_act), "_None")


def adam_step(p, g, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """One step of the trainer's optimizer (``Adam(self.model.parameters(), lr=args.lr)``, reference ``src_1gp/trainer.py:49-50``;
    torch.optim.Adam without amsgrad / maximize, weight decay in the L2 form) on numpy fp32 arrays, in the arithmetic of
    ``csrc/optim.hip`` — fp32 throughout, the bias corrections as ``-expm1(s ln beta)``.  ``step`` = steps taken so far.
    Returns ``(p, m, v)`` after the step.  Pinned against ``torch.optim.Adam`` on the CPU in ``tests/test_host_logic.py``."""
    import numpy as np
    f = np.float32
    p, g, m, v = (np.asarray(t, dtype=f) for t in (p, g, m, v))
    s = f(step + 1)
    if weight_decay:
        g = g + f(weight_decay) * p
    m = m + f(1.0 - beta1) * (g - m)
    v = f(beta2) * v + f(1.0 - beta2) * g * g
    with np.errstate(divide="ignore"):
        bc1 = -np.expm1(s * f(np.log(beta1)), dtype=f) if beta1 > 0 else f(1)
        bc2 = -np.expm1(s * f(np.log(beta2)), dtype=f) if beta2 > 0 else f(1)
    step_size = f(lr) / bc1
    denom = np.sqrt(v) / np.sqrt(bc2) + f(eps)
    return p - step_size * (m / denom), m, v
