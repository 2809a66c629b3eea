"""MI355X drop-in for the reference's graph-layer module (``src_1gp/layer.py``).

Same class names, constructor signatures, parameter names / shapes / initialisation order and
``forward`` surfaces as the reference, so that its string-dispatched configuration
(``exec('self.conv={}(in_dim, out_dim, in_edge_dim)'.format(conv))`` at ``layer.py:227-230,
244-249``), its checkpoints (``trainer.py:113-138``) and its ``seed_torch`` reproducible init
all carry over.  The arithmetic of the message-passing path runs in hand-written gfx950
kernels (``glam_amd/csrc``) reached through the C ABI in ``include/glam_hip.h``; nothing here
falls back to CPU or to an eager re-implementation.

Reference line citations are relative to ``/root/reference``.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch.nn import Parameter, Dropout  # noqa: F401  (names resolved from config strings)
from torch.nn import Sequential, Linear, ReLU, CELU, PReLU, RReLU, LeakyReLU, GRU, Sigmoid  # noqa: F401
from torch.nn.init import kaiming_uniform_, zeros_, ones_  # noqa: F401

from . import ops
from ._lib import GlamHipError


def _ceil4(c):
    return (c + 3) // 4 * 4


def _pad_de(de):
    if de <= 4:
        return 4
    if de <= 8:
        return 8
    raise GlamHipError(f"edge_channels={de} > 8 is outside the compiled kernel table")


# --------------------------------------------------------------------------------------
# MessagePassing surface (PyG base class the reference derives from, layer.py:9)
# --------------------------------------------------------------------------------------
_PROPAGATING = None     # (GraphIndex, target-index tensor) of the propagate() call in flight: lets softmax() reuse its CSR


def softmax(src, index, ptr=None, num_nodes=None):
    """PyG 1.7.2 ``utils.softmax``: per column, softmax over the entries that share ``index`` (max-shifted, denominator
    ``+ 1e-16``).  Segment max / sum run on the HIP segment-reduction kernels (CSR by ``index``)."""
    if ptr is not None:
        raise GlamHipError("softmax(ptr=...) is not used by the reference (it passes ptr=None)")
    N = int(num_nodes) if num_nodes is not None else (int(index.max().item()) + 1 if index.numel() else 0)
    if _PROPAGATING is not None and _PROPAGATING[1] is index and _PROPAGATING[0].N == N:
        gi = _PROPAGATING[0]
    else:
        gi = ops.graph_index(torch.stack([index, index]), N)
    flat = src.reshape(src.size(0), -1)
    m = ops.edge_reduce(flat.detach(), gi, "max").index_select(0, index)       # the shift cancels in the quotient: no gradient
    p = torch.exp(flat - m)
    s_ = ops.edge_reduce(p, gi, "sum").index_select(0, index)
    return (p / (s_ + 1e-16)).view_as(src)


class MessagePassing(torch.nn.Module):
    """PyG 1.7.2 ``MessagePassing`` surface the reference derives from (layer.py:9): constructor (``aggr``, ``flow``,
    ``node_dim``), ``propagate``, ``message``, ``update``.

    ``propagate(edge_index, size=None, **kwargs)`` runs the PyG pipeline — collect (``name_j`` = ``kwargs[name]`` gathered at
    ``edge_index[0]``, ``name_i`` at ``edge_index[1]``, ``edge_index_i/_j``, ``size_i/_j``, everything else passed through) ->
    ``message`` -> aggregate at the targets (HIP CSR segment reduction: ``add`` / ``mean`` / ``max``, no atomics) -> ``update`` —
    for any subclass that defines ``message`` / ``update`` the PyG way.  The reference's own convs never take this op-by-op
    route here: ``TripletMessage`` / ``TripletMessageLight`` override ``propagate`` and hand the whole pipeline to the fused
    kernels (their ``forward`` calls it with the raw inputs)."""

    def __init__(self, aggr="add", flow="source_to_target", node_dim=0, **kwargs):
        super().__init__()
        if flow != "source_to_target":
            raise ValueError("only flow='source_to_target' is supported (the reference never changes it)")
        if aggr not in ("add", "sum", "mean", "max"):
            raise ValueError(f"aggr={aggr!r}: expected 'add', 'mean' or 'max'")
        if node_dim != 0:
            raise ValueError("only node_dim=0 is supported (the reference's convs all use it)")
        self.aggr, self.flow, self.node_dim = aggr, flow, node_dim

    def _collect(self, edge_index, size, kwargs):
        import inspect
        names = [n for n in inspect.signature(self.message).parameters if n not in ("self",)]
        n_nodes = size if isinstance(size, int) else (size[1] if size is not None else None)
        for v in kwargs.values():
            if n_nodes is None and torch.is_tensor(v) and v.dim() >= 1:
                n_nodes = v.size(0)                    # PyG: the node count is the size of the first lifted tensor
                break
        args = {}
        for name in names:
            if name == "edge_index_i":
                args[name] = edge_index[1]
            elif name == "edge_index_j":
                args[name] = edge_index[0]
            elif name in ("size_i", "size_j"):
                args[name] = n_nodes
            elif name.endswith("_i") and name[:-2] in kwargs:
                args[name] = kwargs[name[:-2]].index_select(0, edge_index[1])
            elif name.endswith("_j") and name[:-2] in kwargs:
                args[name] = kwargs[name[:-2]].index_select(0, edge_index[0])
            elif name in kwargs:
                args[name] = kwargs[name]
            else:
                raise TypeError(f"propagate(): message() needs '{name}' but it was not passed")
        return args, n_nodes

    def propagate(self, edge_index, size=None, **kwargs):
        global _PROPAGATING
        ops.require_device(edge_index)
        args, n_nodes = self._collect(edge_index, size, kwargs)
        if n_nodes is None:
            raise GlamHipError("propagate(): cannot infer the node count; pass size=N")
        gi = ops.graph_index(edge_index, n_nodes)
        prev, _PROPAGATING = _PROPAGATING, (gi, args.get("edge_index_i"))
        try:
            msg = self.message(**args)
        finally:
            _PROPAGATING = prev
        out = ops.edge_reduce(msg.reshape(msg.size(0), -1), gi, "sum" if self.aggr == "add" else self.aggr)
        return self.update(out.view((n_nodes,) + tuple(msg.shape[1:])))

    def message(self, x_j):
        return x_j

    def update(self, aggr_out):
        return aggr_out


# --------------------------------------------------------------------------------------
# TripletMessage (layer.py:15-64)
# --------------------------------------------------------------------------------------
class TripletMessage(MessagePassing):
    def __init__(self, node_channels, edge_channels, heads=3, negative_slope=0.2, **kwargs):
        super().__init__(aggr="add", node_dim=0, **kwargs)
        self.node_channels, self.edge_channels = node_channels, edge_channels
        self.heads, self.negative_slope = heads, negative_slope
        # same creation + init order as layer.py:22-34 (uninitialised tensors consume no RNG)
        self.weight_node = Parameter(torch.empty(node_channels, heads * node_channels))
        self.weight_edge = Parameter(torch.empty(edge_channels, heads * node_channels))
        self.weight_triplet_att = Parameter(torch.empty(1, heads, 3 * node_channels))
        self.weight_scale = Parameter(torch.empty(heads * node_channels, node_channels))
        self.bias = Parameter(torch.empty(node_channels))
        self.reset_parameters()

    def reset_parameters(self):
        kaiming_uniform_(self.weight_node)
        kaiming_uniform_(self.weight_edge)
        kaiming_uniform_(self.weight_triplet_att)
        kaiming_uniform_(self.weight_scale)
        zeros_(self.bias)

    def _staged_weights(self):
        """Parameter-only staging (no node/edge data): separable attention weights and the
        4-channel padded layouts the kernels read.  ``logit = a_i[dst] + <edge_attr, M> + a_j[src]``
        with ``a_i = x @ Wa[:, :4]``, ``a_j = x @ Wa[:, 4:]``."""
        H, C, De = self.heads, self.node_channels, self.edge_channels
        Cp, Dp = _ceil4(C), _pad_de(De)
        att = self.weight_triplet_att[0]                                 # [H, 3C]
        att_ij = torch.stack([att[:, :C], att[:, 2 * C:]], dim=-1)      # [H, C, 2]
        Wn = self.weight_node.view(C, H, C)
        Wa = torch.bmm(Wn.permute(1, 0, 2), att_ij)                      # [H, Cin, 2]
        Wa = F.pad(Wa.permute(1, 2, 0), (0, 4 - H)).reshape(C, 8)        # [Cin, (i|j) x 4]
        We = self.weight_edge.view(De, H, C)
        M = torch.bmm(We.permute(1, 0, 2), att[:, C:2 * C].unsqueeze(-1)).squeeze(-1).t()   # [De, H]
        M = F.pad(M, (0, 4 - H, 0, Dp - De))
        if Cp != C or Dp != De:
            Wn = F.pad(Wn, (0, Cp - C))
            We = F.pad(We, (0, Cp - C, 0, 0, 0, Dp - De))
            Ws = F.pad(self.weight_scale.view(H, C, C), (0, 0, 0, Cp - C)).reshape(H * Cp, C)
        else:
            Ws = self.weight_scale
        return Wn.reshape(C, H * Cp), Wa, We.reshape(Dp, H * Cp).contiguous(), M.contiguous(), Ws, Cp, Dp

    def forward(self, x, edge_index, edge_attr, size=None):
        # layer.py:36-40 hands `x @ weight_node` and `edge_attr @ weight_edge` to propagate(); here the two products are part of
        # the fused pipeline, so propagate() receives the raw tensors (it accepts both forms)
        return self.propagate(edge_index, x=x, edge_attr=edge_attr, size=size)

    def propagate(self, edge_index, size=None, x=None, edge_attr=None, **kwargs):
        """``x[N, C]`` with ``edge_attr[E, De]`` (raw inputs): the whole layer in the fused kernels.  ``x[N, H*C]`` with
        ``edge_attr[E, H*C]`` (already multiplied by ``weight_node`` / ``weight_edge``, the tensors the reference's forward passes,
        layer.py:37-40): PyG's collect -> message -> aggregate -> update pipeline on those tensors."""
        if x is None or edge_attr is None or kwargs:
            raise GlamHipError("TripletMessage.propagate(edge_index, x=..., edge_attr=..., size=None)")
        edge_attr = edge_attr.unsqueeze(-1) if edge_attr.dim() == 1 else edge_attr
        HC = self.heads * self.node_channels
        if x.size(1) == HC and edge_attr.size(1) == HC and not (self.heads == 1 and edge_attr.size(1) == self.edge_channels):
            return MessagePassing.propagate(self, edge_index, size=size, x=x, edge_attr=edge_attr)
        if x.size(1) != self.node_channels or edge_attr.size(1) != self.edge_channels:
            raise GlamHipError(f"TripletMessage.propagate: x has {x.size(1)} and edge_attr {edge_attr.size(1)} columns; expected "
                               f"({self.node_channels}, {self.edge_channels}) raw or ({HC}, {HC}) transformed")
        return self._fused(x, edge_index, edge_attr)

    def message(self, x_j, x_i, edge_index_i, edge_attr, size_i):                       # layer.py:42-55
        """Per-edge messages ``alpha * e_ij * x_j`` [E, H, C] from the lifted, already transformed tensors; the logit of
        (edge, head) is the three-part dot product <x_i, att_i> + <e_ij, att_e> + <x_j, att_j> (the concatenation the reference
        forms is never built)."""
        H, C = self.heads, self.node_channels
        xj, xi, e = x_j.view(-1, H, C), x_i.view(-1, H, C), edge_attr.view(-1, H, C)
        att = self.weight_triplet_att[0]                                                # [H, 3C]
        logit = (xi * att[:, :C]).sum(-1) + (e * att[:, C:2 * C]).sum(-1) + (xj * att[:, 2 * C:]).sum(-1)
        alpha = softmax(F.leaky_relu(logit, self.negative_slope), edge_index_i, ptr=None, num_nodes=size_i)
        return alpha.unsqueeze(-1) * e * xj

    def forward_with_identity(self, x, edge_index, edge_attr):
        """``(self(x, ...), identity)`` with ``identity`` = ``x`` handed back through the layer's autograd node where that saves the
        add launch of the skip connection (MessageBlock, layer.py:253-265); ``identity`` is plain ``x`` on every other route."""
        C, De = self.node_channels, self.edge_channels
        if (self.heads <= 4 and x.dim() == 2 and x.size(1) == C and edge_attr.dim() == 2 and edge_attr.size(1) == De
                and ops.fused_layer_supported(C, self.heads, De) and x.is_cuda and torch.is_grad_enabled() and x.requires_grad):
            gi = ops.graph_index(edge_index, x.size(0))
            Cp, Dp = _ceil4(C), _pad_de(De)
            if Dp != De:
                edge_attr = F.pad(edge_attr, (0, Dp - De))
            out, ident = ops.triplet_layer(ops.pad_cols(x, Cp), edge_attr, self.weight_node, self.weight_edge, self.weight_triplet_att,
                                           self.weight_scale, self.bias, gi, self.heads, self.negative_slope, with_identity=True)
            return ops.slice_cols(out, C), ops.slice_cols(ident, C)
        return self(x, edge_index, edge_attr), x

    def _head_groups(self, x, edge_index, edge_attr):
        """``heads > 4`` (the kernels keep at most four heads of a node in a lane group): the layer is a SUM over head groups — every
        head contributes ``aggr_h @ W_scale[h]`` to the update (layer.py:57-61) and nothing else couples heads — so it runs as
        ceil(heads / g) layers of g <= 4 heads on slices of the parameters, the bias added by the first."""
        H, C, De = self.heads, self.node_channels, self.edge_channels
        g = next((k for k in (4, 3, 2, 1) if ops.fused_layer_supported(C, k, De) or ops.wide_layer_supported(C, k, De)), 0)
        if g == 0:
            raise GlamHipError(f"TripletMessage({C}, {De}, heads={H}): width outside the compiled kernel table")
        gi = ops.graph_index(edge_index, x.size(0))
        Cp, Dp = _ceil4(C), _pad_de(De)
        if Dp != edge_attr.size(1):
            edge_attr = F.pad(edge_attr, (0, Dp - edge_attr.size(1)))
        x_p = ops.pad_cols(x, Cp)
        out = None
        for h0 in range(0, H, g):
            h1 = min(h0 + g, H)
            k = h1 - h0
            wn = self.weight_node.view(C, H, C)[:, h0:h1].reshape(C, k * C).contiguous()
            we = self.weight_edge.view(De, H, C)[:, h0:h1].reshape(De, k * C).contiguous()
            att = self.weight_triplet_att[:, h0:h1].contiguous()
            ws = self.weight_scale.view(H, C, C)[h0:h1].reshape(k * C, C).contiguous()
            bias = self.bias if h0 == 0 else torch.zeros_like(self.bias)
            layer_fn = ops.triplet_layer if ops.fused_layer_supported(C, k, De) else ops.triplet_layer_wide
            o = layer_fn(x_p, edge_attr, wn, we, att, ws, bias, gi, k, self.negative_slope)
            out = o if out is None else out + o
        return ops.slice_cols(out, C)

    def _fused(self, x, edge_index, edge_attr):
        if self.heads > 4:
            return self._head_groups(x, edge_index, edge_attr)
        gi = ops.graph_index(edge_index, x.size(0))
        C, De = self.node_channels, self.edge_channels
        Cp, Dp = _ceil4(C), _pad_de(De)
        if Dp != edge_attr.size(1):
            edge_attr = F.pad(edge_attr, (0, Dp - edge_attr.size(1)))
        if ops.fused_layer_supported(C, self.heads, De):
            # whole layer in HIP: staging, MFMA node GEMM, fused aggregate, MFMA update (+ its backward)
            out = ops.triplet_layer(ops.pad_cols(x, Cp), edge_attr, self.weight_node, self.weight_edge, self.weight_triplet_att,
                                    self.weight_scale, self.bias, gi, self.heads, self.negative_slope)
            return ops.slice_cols(out, C)                # pad columns are exactly zero (zero-padded W_scale / bias)
        # wide layers (3 * Cp + 8 > 192, i.e. hid_dim_alpha = 6): fused aggregate kernel between library GEMMs
        x_p = ops.pad_cols(x, Cp)
        if ops.wide_layer_supported(C, self.heads, De):
            out = ops.triplet_layer_wide(x_p, edge_attr, self.weight_node, self.weight_edge, self.weight_triplet_att,
                                         self.weight_scale, self.bias, gi, self.heads, self.negative_slope)
            return ops.slice_cols(out, C)
        Wn, Wa, We, M, Ws, Cp, Dp = ops.scoped_weights(("triplet-derived", id(self.weight_node)), self.weight_node,
                                                       self._staged_wide)
        # every N-deep weight gradient of this path runs on k_wgrad (ops.matmul_tall: the library's heuristics give those
        # products 32x32 tiles, 75 us each); the data-side products stay on the library GEMM
        xw = ops.matmul_tall(x_p, Wn)                                      # layer.py:37
        a_ij = ops.matmul_tall(x_p, Wa)
        aggr = ops.triplet_aggregate(xw, a_ij, edge_attr, We, M, gi, self.heads, Cp, self.negative_slope)
        return ops.matmul_tall(aggr, Ws)[:, :C] + self.bias                # layer.py:57-61

    def _staged_wide(self):
        """``_staged_weights`` with the input rows of ``W_node`` / ``W_a`` and the output columns of ``W_scale`` zero-padded
        to ``Cp`` (the wide path pads ``x`` and slices the output)."""
        Wn, Wa, We, M, Ws, Cp, Dp = self._staged_weights()
        if Cp != self.node_channels:
            Wn, Wa = F.pad(Wn, (0, 0, 0, Cp - self.node_channels)), F.pad(Wa, (0, 0, 0, Cp - self.node_channels))
            Ws = F.pad(Ws, (0, Cp - self.node_channels))
        return Wn.contiguous(), Wa.contiguous(), We, M, Ws.contiguous(), Cp, Dp

    def update(self, aggr_out):                                                          # layer.py:57-61
        return torch.addmm(self.bias, aggr_out.reshape(-1, self.weight_scale.size(0)), self.weight_scale)

    def extra_repr(self):
        return "{node_channels}, {node_channels}, heads={heads}".format(**self.__dict__)


# --------------------------------------------------------------------------------------
# TripletMessageLight (layer.py:67-104)
# --------------------------------------------------------------------------------------
class TripletMessageLight(MessagePassing):
    def __init__(self, node_channels, edge_channels, negative_slope=0.2, **kwargs):
        super().__init__(aggr="add", node_dim=0, **kwargs)
        self.node_channels, self.edge_channels = node_channels, edge_channels
        self.negative_slope = negative_slope
        self.weight_node = Parameter(torch.empty(node_channels, node_channels))
        self.weight_triplet_att = Parameter(torch.empty(1, 2 * node_channels + edge_channels))
        self.bias = Parameter(torch.empty(node_channels))
        self.reset_parameters()

    def reset_parameters(self):
        kaiming_uniform_(self.weight_node)
        kaiming_uniform_(self.weight_triplet_att)
        zeros_(self.bias)

    def forward(self, x, edge_index, edge_attr, size=None):
        return self.propagate(edge_index, x=x, edge_attr=edge_attr, size=size, _raw=True)

    def propagate(self, edge_index, size=None, x=None, edge_attr=None, _raw=False, **kwargs):
        """``_raw=True`` (what ``forward`` passes): ``x`` is the layer input and the whole layer runs in the fused kernels.
        Otherwise ``x`` is ``x @ weight_node`` as in the reference's forward (layer.py:84-86; both forms have C columns, hence
        the flag) and PyG's collect -> message -> aggregate -> update pipeline runs on it."""
        if x is None or edge_attr is None or kwargs:
            raise GlamHipError("TripletMessageLight.propagate(edge_index, x=..., edge_attr=..., size=None)")
        edge_attr = edge_attr.unsqueeze(-1) if edge_attr.dim() == 1 else edge_attr
        if not _raw:
            return MessagePassing.propagate(self, edge_index, size=size, x=x, edge_attr=edge_attr)
        return self._fused(x, edge_index, edge_attr)

    def message(self, x_j, x_i, edge_index_i, edge_attr, size_i):                       # layer.py:88-97
        C, De = self.node_channels, self.edge_channels
        att = self.weight_triplet_att[0]                                                # [2C + De]: x_i | e_ij | x_j
        logit = x_i @ att[:C] + edge_attr @ att[C:C + De] + x_j @ att[C + De:]
        alpha = softmax(F.leaky_relu(logit, self.negative_slope), edge_index_i, ptr=None, num_nodes=size_i)
        return alpha.unsqueeze(-1) * x_j

    def _fused(self, x, edge_index, edge_attr):
        C, De = self.node_channels, self.edge_channels
        Cp, Dp = _ceil4(C), _pad_de(De)
        gi = ops.graph_index(edge_index, x.size(0))

        def derived():      # parameter-only staging, shared by the message_steps applications of the block (ops.weight_scope)
            att = self.weight_triplet_att[0]
            att_ij = torch.stack([att[:C], att[C + De:]], dim=-1)                       # [C, 2]
            Wa = torch.matmul(self.weight_node, att_ij)                                 # [Cin, 2]
            Wa = F.pad(Wa.unsqueeze(-1), (0, 3)).reshape(C, 8)
            M = F.pad(att[C:C + De].unsqueeze(-1), (0, 3, 0, Dp - De)).contiguous()     # [Dp, 4]
            Wn = F.pad(self.weight_node, (0, Cp - C)) if Cp != C else self.weight_node
            wt = torch.cat([Wn, Wa], dim=1)
            if Cp != C:
                wt = F.pad(wt, (0, 0, 0, Cp - C))
            return Wn, Wa, M, wt

        Wn, Wa, M, wt = ops.scoped_weights(("light-derived", id(self.weight_node)), self.weight_node, derived)
        if ops.linear_split_supported(Cp, Cp + 8):
            # node GEMM + separable attention columns in one MFMA launch: [xw | a_i a_j] = x @ [W_node | Wa]
            x_p = F.pad(x, (0, Cp - C)) if Cp != C else x
            xw, a_ij = ops.linear_split(x_p, wt, Cp)                                # layer.py:84
        else:
            xw = torch.matmul(x, Wn)
            a_ij = torch.matmul(x, Wa)
        if Dp != De:
            edge_attr = F.pad(edge_attr, (0, Dp - De))
        aggr = ops.light_aggregate(xw, a_ij, edge_attr, M, gi, Cp, self.negative_slope)
        return self.update(aggr[:, :C] if Cp != C else aggr)

    def update(self, aggr_out):                                                     # layer.py:99-101
        return aggr_out + self.bias

    def extra_repr(self):
        return "{node_channels}, {node_channels}".format(**self.__dict__)


class _None(torch.nn.Module):  # placeholder for no norm / dropout / activation (layer.py:107-112)
    def __init__(self, **params):
        super().__init__()

    def forward(self, x, batch=None):
        return x


# --------------------------------------------------------------------------------------
# PyG convolutions the search space can select (layer.py:115-158)
# --------------------------------------------------------------------------------------
class NNConv(MessagePassing):
    """PyG 1.7.2 ``NNConv``: ``x_i' = x_i @ root + aggr_j(x_j @ nn(e_ij).view(in,out)) + bias``."""

    def __init__(self, in_channels, out_channels, nn, aggr="add", root_weight=True, bias=True, **kwargs):
        super().__init__(aggr=aggr, node_dim=0, **kwargs)
        self.in_channels, self.out_channels, self.nn = in_channels, out_channels, nn
        self.root = Parameter(torch.empty(in_channels, out_channels)) if root_weight else None
        self.bias = Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        for m in self.nn.modules():
            if m is not self.nn and hasattr(m, "reset_parameters"):
                m.reset_parameters()
        if self.root is not None:
            bound = 1.0 / math.sqrt(self.root.size(0))
            with torch.no_grad():
                self.root.uniform_(-bound, bound)
        if self.bias is not None:
            zeros_(self.bias)

    def forward_with_identity(self, x, edge_index, edge_attr):
        """``(self(x, ...), identity)`` with ``identity`` = ``x`` handed back through the relation-sum node where the layer runs as one
        stacked GEMM (one-hot bonds, root term as the self slot): the skip connection's gradient is added inside that node's backward
        launch; plain ``x`` on every other route."""
        return self.forward(x, edge_index, edge_attr, with_identity=True)

    def forward(self, x, edge_index, edge_attr, size=None, with_identity=False):
        out = self._forward(x, edge_index, edge_attr, with_identity)
        if with_identity:
            return out if isinstance(out, tuple) else (out, x)
        return out

    def _forward(self, x, edge_index, edge_attr, with_identity):
        gi = ops.graph_index(edge_index, x.size(0))
        De = edge_attr.size(1)
        if De <= 8 and not edge_attr.requires_grad and ops.rows_are_one_hot(edge_attr):
            # Bond features are one-hot (src_1gp/dataset.py:82): nn(e_ij) takes only De distinct values, so the layer is a
            # De-relation R-GCN: per-relation neighbour sums (one HIP kernel) followed by ONE [N, De*C] x [De*C, C] GEMM,
            # instead of the reference's [E, C*C] per-edge weight tensor (612 MB at B=1024).
            Dp = _pad_de(De)
            def relation_weights():     # parameter-only: shared by the message_steps applications (ops.weight_scope)
                w = ops.relation_mlp(self.nn, De)                                        # nn(eye(De)): [De, in*out]
                return w.view(De * self.in_channels, self.out_channels)

            w_rel = ops.scoped_weights(("nnconv-rel", id(self), De), self, relation_weights)
            ea = F.pad(edge_attr, (0, Dp - De)) if Dp != De else edge_attr
            if self.aggr not in ("mean", "add", "sum"):
                raise GlamHipError("NNConv: only aggr in {'mean', 'add'} is supported")
            C = self.in_channels
            if self.root is not None and ops.self_slot_supported(Dp, C) and (Dp + 1) * C + 1 <= 320 and self.out_channels % 4 == 0:
                # the root term x_i @ root as one more relation (the node's own row in slot Dp of the relation sums): the whole
                # layer is ONE [N, (Dp+1) C] x [(Dp+1) C, out] GEMM with the bias in its epilogue, and one k_wgrad launch back
                def stacked():
                    w = w_rel if Dp == De else F.pad(w_rel.view(De, C, -1), (0, 0, 0, 0, 0, Dp - De)).reshape(Dp * C, -1)
                    return torch.cat([w, self.root], dim=0).contiguous()
                w_all = ops.scoped_weights(("nnconv-stack", id(self), De), self, stacked)
                if with_identity:
                    S, ident = ops.edge_weighted_sum(x, ea, gi, mean=(self.aggr == "mean"), self_slot=True, with_identity=True)
                    return ops.matmul_tall(S.view(x.size(0), (Dp + 1) * C), w_all, self.bias), ident
                S = ops.edge_weighted_sum(x, ea, gi, mean=(self.aggr == "mean"), self_slot=True)      # [N, Dp + 1, in]
                return ops.matmul_tall(S.view(x.size(0), (Dp + 1) * C), w_all, self.bias)
            S = ops.edge_weighted_sum(x, ea, gi, mean=(self.aggr == "mean"))           # [N, Dp, in]
            out = ops.matmul_tall(S[:, :De].reshape(x.size(0), De * self.in_channels), w_rel)
        else:
            weight = self.nn(edge_attr).view(-1, self.in_channels, self.out_channels)
            msg = torch.bmm(x.index_select(0, edge_index[0]).unsqueeze(1), weight).squeeze(1)
            out = ops.edge_reduce(msg, gi, self.aggr)
        if self.root is not None:
            out = out + ops.matmul_tall(x, self.root)
        if self.bias is not None:
            out = out + self.bias
        return out


def _glorot(t):
    stdv = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-stdv, stdv)


def _with_self_loops(gi, edge_index, n, keep_existing):
    """``edge_index`` + one self loop per node (GCN keeps existing loops' weight slot, GAT drops
    them first); staged once per edge list."""
    key = ("loops", keep_existing)
    hit = gi.__dict__.setdefault("_derived", {}).get(key)
    if hit is None:
        mask = edge_index[0] != edge_index[1]
        loop = torch.arange(n, dtype=edge_index.dtype, device=edge_index.device).unsqueeze(0).repeat(2, 1)
        ei = torch.cat([edge_index[:, mask], loop], dim=1).contiguous()
        hit = (ei, ops.GraphIndex(ei, n), mask)
        gi._derived[key] = hit
    return hit


class GCNConv(MessagePassing):
    def __init__(self, in_channels, out_channels, bias=True, **kwargs):
        super().__init__(aggr="add", node_dim=0, **kwargs)
        self.weight = Parameter(torch.empty(in_channels, out_channels))
        self.bias = Parameter(torch.empty(out_channels)) if bias else None
        _glorot(self.weight)
        if self.bias is not None:
            zeros_(self.bias)

    @staticmethod
    def _norm(ei, gi, w):
        deg = ops.edge_reduce(w, gi, "sum")
        dis = deg.pow(-0.5)
        dis = dis.masked_fill(dis == float("inf"), 0)
        return dis[ei[0]] * w * dis[ei[1]]

    def forward_with_identity(self, x, edge_index):
        """``(self(x, edge_index, add_bias=False), identity)`` with ``identity`` = ``x`` handed back through the node of the ``x @ W``
        product where that saves the add launch of the skip connection around the conv (MessageBlock, layer.py:253-265)."""
        return self.forward(x, edge_index, add_bias=False, with_identity=True)

    def forward(self, x, edge_index, edge_weight=None, add_bias=True, with_identity=False):
        n = x.size(0)
        gi0 = ops.graph_index(edge_index, n)
        ei, gi, mask = _with_self_loops(gi0, edge_index, n, True)
        if edge_weight is None:        # the reference's call (layer.py:148): the normalisation is a function of the edge list
            cache = gi.__dict__.setdefault("_derived", {})
            norm = cache.get("gcn_norm")
            if norm is None:
                norm = cache["gcn_norm"] = self._norm(ei, gi, x.new_ones(ei.size(1))).view(-1, 1).contiguous()
        else:  # add_remaining_self_loops: existing loops keep their weight
            loop_w = edge_weight.new_ones(n)
            inv = ~mask
            loop_w[edge_index[0][inv]] = edge_weight[inv]
            norm = self._norm(ei, gi, torch.cat([edge_weight[mask], loop_w])).view(-1, 1)
        ident = x
        if with_identity:
            xw, ident = ops.matmul_tall(x, self.weight, with_identity=True)
        else:
            xw = ops.matmul_tall(x, self.weight)
        if norm.requires_grad:         # learnable edge weights: per-edge messages, so that autograd reaches them
            out = ops.edge_reduce(norm * xw.index_select(0, ei[0]), gi, "sum")
        else:                          # one gather-scale-sum kernel per direction (K = 1 relation)
            out = ops.edge_weighted_sum(xw, norm, gi).view(n, xw.size(1))
        out = out if (self.bias is None or not add_bias) else out + self.bias
        return (out, ident) if with_identity else out


class GATConv(MessagePassing):
    """PyG 1.7.2 ``GATConv(in, out)`` with its defaults (heads=1, concat, self loops).  The
    attention softmax-aggregate is the same fused kernel as TripletMessageLight."""

    def __init__(self, in_channels, out_channels, heads=1, negative_slope=0.2, bias=True, **kwargs):
        super().__init__(aggr="add", node_dim=0, **kwargs)
        if heads != 1:
            raise GlamHipError("GATConv: only heads=1 (the reference's configuration) is supported")
        self.heads, self.out_channels, self.negative_slope = heads, out_channels, negative_slope
        self.lin_l = Linear(in_channels, heads * out_channels, bias=False)
        self.lin_r = self.lin_l
        self.att_l = Parameter(torch.empty(1, heads, out_channels))
        self.att_r = Parameter(torch.empty(1, heads, out_channels))
        self.bias = Parameter(torch.empty(heads * out_channels)) if bias else None
        _glorot(self.lin_l.weight)
        _glorot(self.att_l)
        _glorot(self.att_r)
        if self.bias is not None:
            zeros_(self.bias)

    def forward(self, x, edge_index, add_bias=True):
        n, C = x.size(0), self.out_channels
        Cp = _ceil4(C)
        gi0 = ops.graph_index(edge_index, n)
        ei, gi, _ = _with_self_loops(gi0, edge_index, n, False)
        xl = ops.linear(x, self.lin_l.weight, self.lin_l.bias)    # MFMA GEMM + k_wgrad (library fallback outside their table)
        a_r = (xl * self.att_r.view(1, C)).sum(-1, keepdim=True)   # target side ("alpha_i")
        a_l = (xl * self.att_l.view(1, C)).sum(-1, keepdim=True)   # source side ("alpha_j")
        a_ij = torch.cat([F.pad(a_r, (0, 3)), F.pad(a_l, (0, 3))], dim=1)
        if Cp != C:
            xl = F.pad(xl, (0, Cp - C))
        zeros_e = x.new_zeros(ei.size(1), 4)
        aggr = ops.light_aggregate(xl, a_ij, zeros_e, x.new_zeros(4, 4), gi, Cp, self.negative_slope)
        out = aggr[:, :C] if Cp != C else aggr
        return out if (self.bias is None or not add_bias) else out + self.bias


class _NNConv(torch.nn.Module):  # layer.py:115-122
    def __init__(self, in_dim, out_dim, edge_in_dim):
        super().__init__()
        nn = Sequential(Linear(edge_in_dim, 32), ReLU(), Linear(32, in_dim * out_dim))
        self.conv = NNConv(in_dim, out_dim, nn, aggr="mean")

    def forward(self, x, edge_index, edge_attr):
        return self.conv(x, edge_index, edge_attr)

    def forward_with_identity(self, x, edge_index, edge_attr):
        return self.conv.forward_with_identity(x, edge_index, edge_attr)


class _TripletMessage(torch.nn.Module):  # layer.py:125-131
    def __init__(self, in_dim, out_dim, edge_in_dim):
        super().__init__()
        self.conv = TripletMessage(in_dim, edge_in_dim)  # in_dim == out_dim

    def forward(self, x, edge_index, edge_attr):
        return self.conv(x, edge_index, edge_attr)

    def forward_with_identity(self, x, edge_index, edge_attr):
        return self.conv.forward_with_identity(x, edge_index, edge_attr)


class _TripletMessageLight(torch.nn.Module):  # layer.py:134-140
    def __init__(self, in_dim, out_dim, edge_in_dim):
        super().__init__()
        self.conv = TripletMessageLight(in_dim, edge_in_dim)

    def forward(self, x, edge_index, edge_attr):
        return self.conv(x, edge_index, edge_attr)


class _GCNConv(torch.nn.Module):  # layer.py:143-149
    def __init__(self, in_dim, out_dim, edge_in_dim):
        super().__init__()
        self.conv = GCNConv(in_dim, out_dim)

    def forward(self, x, edge_index, edge_attr):
        return self.conv(x, edge_index)


class _GATConv(torch.nn.Module):  # layer.py:152-158
    def __init__(self, in_dim, out_dim, edge_in_dim):
        super().__init__()
        self.conv = GATConv(in_dim, out_dim)

    def forward(self, x, edge_index, edge_attr):
        return self.conv(x, edge_index)


# --------------------------------------------------------------------------------------
# graph norms (PyG classes wrapped at layer.py:161-194); segment statistics on the HIP pools
# --------------------------------------------------------------------------------------
class BatchNorm(torch.nn.Module):
    def __init__(self, in_channels, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.module = torch.nn.BatchNorm1d(in_channels, eps, momentum, affine, track_running_stats)

    def forward(self, x):
        return self.module(x)


class LayerNorm(torch.nn.Module):
    """PyG graph LayerNorm: statistics over all nodes x channels of each graph."""

    def __init__(self, in_channels, eps=1e-5, affine=True):
        super().__init__()
        self.in_channels, self.eps = in_channels, eps
        self.weight = Parameter(torch.ones(in_channels)) if affine else None
        self.bias = Parameter(torch.zeros(in_channels)) if affine else None

    def forward(self, x, batch=None, with_identity=False):
        ident = x
        if batch is None:
            x = x - x.mean()
            out = x / (x.std(unbiased=False) + self.eps)
        elif with_identity:
            out, ident = ops.graph_standardize(x, ops.segment_ptr(batch), self.eps, with_identity=True)
        else:
            out = ops.graph_standardize(x, ops.segment_ptr(batch), self.eps)
        if self.weight is not None:
            out = out * self.weight + self.bias
        return (out, ident) if with_identity else out


class PairNorm(torch.nn.Module):
    def __init__(self, scale=1.0, scale_individually=False, eps=1e-5):
        super().__init__()
        if scale_individually:
            raise GlamHipError("PairNorm(scale_individually=True) is not used by the reference")
        self.scale, self.eps = scale, eps

    def forward(self, x, batch=None, with_identity=False, drop_p=0.0):
        if batch is None:
            xc = x - x.mean(dim=0, keepdim=True)
            out = self.scale * xc / (self.eps + xc.pow(2).sum(-1).mean()).sqrt()
            return (out, x) if with_identity else out
        return ops.pair_norm(x, ops.segment_ptr(batch), self.scale, self.eps, with_identity=with_identity, drop_p=drop_p)

    def drop_supported(self, x, batch):
        """The training-mode Dropout behind this norm can come from the norm's own launch (``forward(..., drop_p=p)``)."""
        return batch is not None and ops.graph_norm_drop_supported(x, ops.segment_ptr(batch))


class GraphSizeNorm(torch.nn.Module):
    def forward(self, x, batch=None):
        if batch is None:
            return x * (x.size(0) ** -0.5)
        sp = ops.segment_ptr(batch)
        inv = (sp.ptr[1:] - sp.ptr[:-1]).to(x.dtype).pow(-0.5)
        return x * inv.index_select(0, batch).view(-1, 1)


class _BatchNorm(torch.nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.norm = BatchNorm(in_channels)

    def forward(self, x, batch=None):
        return self.norm(x)


class _LayerNorm(torch.nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.norm = LayerNorm(in_channels)

    def forward(self, x, batch=None, with_identity=False):
        return self.norm(x, batch, with_identity) if with_identity else self.norm(x, batch)


class _PairNorm(torch.nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.norm = PairNorm()

    def forward(self, x, batch=None, with_identity=False, drop_p=0.0):
        if drop_p > 0:
            return self.norm(x, batch, with_identity, drop_p)
        return self.norm(x, batch, with_identity) if with_identity else self.norm(x, batch)


class _GraphSizeNorm(torch.nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.norm = GraphSizeNorm()

    def forward(self, x, batch=None):
        return self.norm(x)   # the reference drops ``batch`` here (layer.py:193-194)


# --------------------------------------------------------------------------------------
# readouts (layer.py:197-220, model.py:41)
# --------------------------------------------------------------------------------------
def global_add_pool(x, batch, size=None):
    return ops.segment_pool(x, ops.segment_ptr(batch, size), "sum")


def global_mean_pool(x, batch, size=None):
    return ops.segment_pool(x, ops.segment_ptr(batch, size), "mean")


def global_max_pool(x, batch, size=None):
    return ops.segment_pool(x, ops.segment_ptr(batch, size), "max")


def global_sort_pool(x, batch, k):
    if k > 8:
        raise GlamHipError("global_sort_pool: k > 8 is outside the compiled kernel table")
    return ops.pool5(x, ops.segment_ptr(batch), k)[:, 2 * x.size(1):]


class GlobalPool5(torch.nn.Module):
    def __init__(self, **params):
        super().__init__()

    def forward(self, x, batch, num_graphs=None):
        return ops.pool5(x, ops.segment_ptr(batch, num_graphs), 3)   # mean | add | sort-pool(k=3)


class GlobalAttention(torch.nn.Module):
    def __init__(self, gate_nn, nn=None):
        super().__init__()
        self.gate_nn, self.nn = gate_nn, nn

    def forward(self, x, batch, size=None):
        x = x.unsqueeze(-1) if x.dim() == 1 else x
        sp = ops.segment_ptr(batch, size)
        # plain Linear modules (the reference's GlobalLAPool): route them through ops.linear so the N-deep weight gradients
        # run on k_wgrad; anything else (Sequential gates) is called as is
        lin = lambda m, t: ops.linear(t, m.weight, m.bias) if type(m) is Linear else m(t)
        gate = lin(self.gate_nn, x).view(-1)
        v = lin(self.nn, x) if self.nn is not None else x
        return ops.segment_attention(gate, v, sp)


class GlobalLAPool(torch.nn.Module):
    def __init__(self, in_channels, **params):
        super().__init__()
        gated_nn = Linear(in_channels, 1)
        nn = Linear(in_channels, 2 * in_channels)
        self.pool = GlobalAttention(gate_nn=gated_nn, nn=nn)

    def forward(self, x, batch, num_graphs=None):
        return self.pool(x, batch, num_graphs)


class Set2Set(torch.nn.Module):
    """PyG ``Set2Set(in_channels, processing_steps)`` (model.py:41).  The LSTM parameters live in a
    ``torch.nn.LSTM`` (checkpoint keys ``lstm.*``); the single-step cell is evaluated explicitly."""

    def __init__(self, in_channels, processing_steps, num_layers=1, **params):
        super().__init__()
        if num_layers != 1:
            raise GlamHipError("Set2Set: num_layers != 1 is not used by the reference")
        self.in_channels, self.out_channels = in_channels, 2 * in_channels
        self.processing_steps, self.num_layers = processing_steps, num_layers
        self.lstm = torch.nn.LSTM(self.out_channels, self.in_channels, num_layers)

    def forward(self, x, batch, num_graphs=None):
        sp = ops.segment_ptr(batch, num_graphs)
        B, C = sp.B, self.in_channels
        h, c = x.new_zeros(B, C), x.new_zeros(B, C)
        q_star = x.new_zeros(B, 2 * C)
        L = self.lstm
        Cp = _ceil4(C)
        fused = ops.query_attention_supported(Cp)
        if fused:                                    # zero-padded rows (the previous block's padded output by reference)
            x_p = ops.pad_cols(x, Cp)
            bias = L.bias_ih_l0 + L.bias_hh_l0
        for _ in range(self.processing_steps):
            if fused:
                # two GEMMs, one gate kernel, one attention-read kernel per step (and per direction)
                gates = torch.addmm(torch.addmm(bias, q_star, L.weight_ih_l0.t()), h, L.weight_hh_l0.t())
                h, c = ops.lstm_cell(gates, c)
                r = ops.query_attention(x_p, h if Cp == C else F.pad(h, (0, Cp - C)), sp)
                q_star = torch.cat([h, r if Cp == C else r[:, :C]], dim=-1)
                continue
            gates = F.linear(q_star, L.weight_ih_l0, L.bias_ih_l0) + F.linear(h, L.weight_hh_l0, L.bias_hh_l0)
            i, f, g, o = gates.chunk(4, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            e = (x * h.index_select(0, batch)).sum(dim=-1)
            r = ops.segment_attention(e, x, sp)
            q_star = torch.cat([h, r], dim=-1)
        return q_star


# --------------------------------------------------------------------------------------
# blocks (layer.py:223-267)
# --------------------------------------------------------------------------------------
def _build(expr, **local):
    """The reference resolves module choices by ``exec`` of config strings inside this namespace
    (layer.py:227-230, 244-249); same mechanism, same names."""
    return eval(expr, globals(), local)  # noqa: S307 - config strings, exactly as the reference


def _act(name):
    name = name[:-2] if name.endswith("()") else name
    return _build(name + "()")


def _apply_dropout(mod, x):
    """``mod(x)`` for the block's dropout slot.  A training-mode ``torch.nn.Dropout`` runs on the device-side Philox stream
    (``ops.dropout``: hipGraph-safe, no mask tensor) — or costs nothing when the kernel that produced ``x`` already wrote the
    dropped twin (``ops.take_dropped``)."""
    if type(mod) is Dropout and mod.training and mod.p > 0 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2:
        if mod.p >= 1:
            return torch.zeros_like(x)
        base = ops.padded_base(x)
        if base is not None:       # odd hidden widths: the zero-padded rows are what flows (0 stays 0), x is their [N, C] view
            twin = ops.take_dropped(base, mod.p)
            return ops.slice_cols(twin if twin is not None else ops.dropout(base, mod.p), x.size(1))
        twin = ops.take_dropped(x, mod.p)
        return twin if twin is not None else ops.dropout(x, mod.p)
    return mod(x)


_ZERO_KEEPING = (ReLU, LeakyReLU, CELU, RReLU)      # act(0) = 0: applied to zero-padded rows, the pad columns stay zero


def _apply_act(mod, x, next_dropout=0.0):
    """``mod(x)`` for an activation slot; training-mode RReLU (the reference's default, model.py:31) draws its slopes from the
    device-side Philox stream.  ``next_dropout`` = p of a training-mode ``Dropout(p)`` that is applied to the result next (nothing in
    between): the RReLU launch then writes the dropped twin too (``ops.take_dropped`` hands it to that dropout, whose own launch —
    and its backward — disappear)."""
    if type(mod) in _ZERO_KEEPING and x.is_cuda and x.dim() == 2:
        base = ops.padded_base(x)
        if base is not None:       # odd hidden widths: no pad / slice copies around the activation
            return ops.slice_cols(_apply_act(mod, base, next_dropout), x.size(1))
    if type(mod) is RReLU and mod.training and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2:
        return ops.rrelu(x, mod.lower, mod.upper, drop_p=float(next_dropout))
    return mod(x)


def flat_then_head(flat, head, x, batch=None):
    """``head(flat(x))`` for two LinearBlocks in a row (``mol_flat`` -> ``lin_out1``, src_1gp/model.py:60-61).  In the reference's default
    configuration in training mode — ``flat`` ends in RReLU, ``head`` is Dropout(p) + a linear with a handful of outputs and nothing else —
    the head reads ``flat``'s PRE-activation and applies both itself (``ops.rrelu_dropout_linear_narrow``): the RReLU launch, the dropped
    twin and their two backward launches disappear.  Every other configuration: the two calls as written."""
    a, d = flat.act, head.dropout
    if (isinstance(flat, LinearBlock) and isinstance(head, LinearBlock) and type(a) is RReLU and a.training and 0 < a.lower <= a.upper
            and isinstance(head.norm, _None) and isinstance(head.act, _None) and x.is_cuda and x.dim() == 2
            and (isinstance(d, _None) or (type(d) is Dropout and (not d.training or 0 <= d.p < 1)))):
        p = float(d.p) if (type(d) is Dropout and d.training) else 0.0
        h = _apply_dropout(flat.dropout, flat.norm(x, batch))
        pre = ops.linear(h, flat.linear.weight, flat.linear.bias)
        y = ops.rrelu_dropout_linear_narrow(pre, head.linear.weight, head.linear.bias, a.lower, a.upper, p) if pre.dim() == 2 else None
        if y is not None:
            return y
        return head(_apply_act(a, pre, following_dropout(head)))
    return head(flat(x, batch, next_dropout=following_dropout(head)))


def first_node_spec(block, n_rows, edge_index, edge_attr):
    """``ops.first_node_spec`` of ``block``'s TripletMessage when the output of the LinearBlock in front of it reaches that conv unchanged
    (no norm; a dropout slot that is empty, inactive, or the training-mode Dropout whose mask that LinearBlock's launch draws:
    ``following_dropout``), else None — the hint for ``LinearBlock.forward(next_node=...)``."""
    conv = block.conv.conv if isinstance(getattr(block, "conv", None), _TripletMessage) else None
    if conv is None or not isinstance(block.norm, _None) or not (isinstance(block.dropout, _None) or type(block.dropout) is Dropout):
        return None
    d = block.dropout
    if type(d) is Dropout and d.training and not (0.0 < d.p < 1.0):
        return None
    return ops.first_node_spec(conv, n_rows, edge_index, edge_attr)


def following_dropout(block):
    """p of the training-mode ``Dropout(p)`` that ``block`` (a LinearBlock / MessageBlock) applies to its input first, 0.0 when
    something else touches the input before it (a norm) or there is none: the hint for the producer's activation launch."""
    d = getattr(block, "dropout", None)
    if type(d) is Dropout and d.training and 0.0 < d.p < 1.0 and isinstance(getattr(block, "norm", None), _None):
        return float(d.p)
    return 0.0


def _prestage_items(lin_block, block, x, edge_attr):
    """(triplet, images) of one tower for ``ops.prestage``: what the first applications of ``lin_block`` (the input LinearBlock) and
    ``block`` (the MessageBlock) would each build with a launch of their own on their default routes."""
    triplet, images, gru_pre = None, [], []
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2):
        return triplet, images, gru_pre
    lin = getattr(lin_block, "linear", None)
    if lin is not None:
        M, Kw = lin.weight.shape
        Kx = _ceil4(x.size(1))
        if x.size(1) == Kw and M % 4 == 0 and ops.linear_supported(Kx, M) and not (Kx != Kw and x.requires_grad):
            images.append(("fwd", ("lin", id(lin.weight)), lin.weight, lin.weight, Kw, 1, Kw, M, Kx))
    conv = getattr(block, "conv", None)
    tm = getattr(conv, "conv", None) if isinstance(conv, _TripletMessage) else None
    if type(tm) is TripletMessage and tm.heads <= 4 and edge_attr is not None and edge_attr.dim() == 2 \
            and edge_attr.size(1) == tm.edge_channels and ops.fused_layer_supported(tm.node_channels, tm.heads, tm.edge_channels):
        triplet = (tm.weight_node, tm.weight_edge, tm.weight_triplet_att, tm.weight_scale, tm.bias, tm.heads, _pad_de(tm.edge_channels))
    gru = getattr(block, "gru", None)
    if gru is not None:
        w_ih, w_hh = gru.weight_ih_l0, gru.weight_hh_l0
        C = w_ih.size(1)
        if C % 4 == 0 and w_ih.shape == (3 * C, C) and ops.gru_images_plain(x.size(0), C):
            images += [("fwd", ("lin", id(w_ih)), w_ih, w_ih, C, 1, C, 3 * C, C), ("fwd", ("lin", id(w_hh)), w_hh, w_hh, C, 1, C, 3 * C, C),
                       ("bwd", ("lin", id(w_ih)), w_ih, w_ih, C, 0, 3 * C, C, 3 * C), ("bwd", ("lin", id(w_hh)), w_hh, w_hh, C, 0, 3 * C, C, 3 * C)]
        elif C % 4 == 0 and w_ih.shape == (3 * C, C) and ops.gru_images_pre(x.size(0), C):
            gru_pre.append((w_ih, w_hh, C))
    return triplet, images, gru_pre


def prestage_pass(*towers):
    """The weight re-layouts of a model pass from as few launches as possible (``ops.prestage``: one TripletMessage staging + six GEMM
    images per launch) instead of one launch per module at its first use.  ``towers``: ``(input LinearBlock, MessageBlock, x,
    edge_attr)`` per graph tower.  The entries land in the pass's ``weight_scope`` under the keys the ops look them up with; only the
    default routes are anticipated, anything else is built by its op as before."""
    built, triplet, images, pres = 0, None, [], []
    for tower in towers:
        t, im, gp = _prestage_items(*tower)
        if (t is not None and triplet is not None) or len(images) + len(im) + 4 * (len(pres) + len(gp)) > 6:
            built += ops.prestage(triplet, images, pres)
            triplet, images, pres = None, [], []
        triplet = t if t is not None else triplet
        images += im
        pres += gp
    if triplet is not None or images or pres:
        built += ops.prestage(triplet, images, pres)
    return built


class LinearBlock(torch.nn.Module):
    def __init__(self, in_dim=32, out_dim=64, norm="_None", dropout="_None()", act="ReLU()"):
        super().__init__()
        self.norm = _build("{}(in_channels=in_dim)".format(norm), in_dim=in_dim)
        self.dropout = _build(dropout)
        self.linear = Linear(in_dim, out_dim)
        self.act = _act(act)

    def forward(self, x, batch=None, next_dropout=0.0, next_node=None):
        """``next_dropout``: p of the training-mode Dropout the block behind applies to this output first (``following_dropout``);
        ``next_node``: that block is a MessageBlock whose TripletMessage reads this output (``first_node_spec``) — the product's launch
        then also writes the dropped twin / the TripletMessage's node product where its kernel can."""
        x = self.norm(x, batch)
        x = _apply_dropout(self.dropout, x)
        a = self.act
        if type(a) is ReLU or type(a) is LeakyReLU or (type(a) is RReLU and not a.training):
            # deterministic (an RReLU in eval mode is a LeakyReLU of its mean slope): the dense linear's epilogue applies it (readout MLP)
            y = ops.linear_act(x, self.linear.weight, self.linear.bias, "relu" if type(a) is ReLU else "leaky",
                               (a.lower + a.upper) / 2 if type(a) is RReLU else getattr(a, "negative_slope", 0.0))
            if y is not None:
                return y
            if type(a) is ReLU:                            # ... or the tall product's (the input embeddings)
                y = ops.linear_relu(x, self.linear.weight, self.linear.bias, node=next_node)
                if y is not None:
                    return y
        if type(a) is RReLU and a.training and 0 < a.lower <= a.upper and x.is_cuda and x.dim() == 2:
            # training-mode RReLU: in the tall product's epilogue too (the input embeddings)
            y = ops.linear_rrelu(x, self.linear.weight, self.linear.bias, a.lower, a.upper, float(next_dropout), node=next_node)
            if y is not None:
                return y
        x = ops.linear(x, self.linear.weight, self.linear.bias)
        return _apply_act(self.act, x, next_dropout)


class MessageBlock(torch.nn.Module):
    def __init__(self, in_dim=32, out_dim=64, in_edge_dim=13, norm="_None", dropout="Dropout(0.2)",
                 conv="_NNConv", act="ReLU()", res=True):
        super().__init__()
        self.norm = _build("{}(in_channels=in_dim)".format(norm), in_dim=in_dim)
        self.dropout = _build(dropout)
        self.conv = _build("{}(in_dim, out_dim, in_edge_dim)".format(conv), in_dim=in_dim, out_dim=out_dim,
                           in_edge_dim=in_edge_dim)
        self.gru = GRU(in_dim, out_dim)
        if conv in ["_GCNConv", "_GATConv"]:
            self.gru = None
        self.act = _act(act)
        self.res = res

    def _gru_step(self, x, h):
        """One step of ``self.gru`` (seq_len 1, layer.py:262) on its own parameters: two MFMA gate GEMMs + fused
        gate kernel."""
        g = self.gru
        return ops.gru_step(x, h, g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0)

    def _norm_drop_p(self, x, batch):
        """p when this block's training-mode Dropout can come out of its norm's launch (a PairNorm over molecule-sized graphs), else 0."""
        d = self.dropout
        if (type(d) is Dropout and d.training and 0.0 < d.p < 1.0 and isinstance(self.norm, _PairNorm) and ops.NORM_DROP
                and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and self.norm.norm.drop_supported(x, batch)):
            return float(d.p)
        return 0.0

    def _fusable_act(self):
        """(code, slope, rng) when ``self.act`` is one the tail kernels apply themselves, else None.  ``rng`` is
        ``(rr_lower, rr_upper, drop_p)`` in training mode when the kernel has random numbers to draw: RReLU slopes, and — when
        nothing sits between this block's output and the next application's dropout (``norm`` is ``_None``) — the Dropout mask
        of the next message step (layer.py:255-256), written as a second output."""
        a = self.act
        d = self.dropout
        drop_p = float(d.p) if (type(d) is Dropout and self.training and 0 < d.p < 1 and isinstance(self.norm, _None)) else 0.0
        rng = (1.0, 1.0, drop_p) if drop_p > 0 else None
        if isinstance(a, _None):
            return "none", 0.0, rng
        if type(a) is torch.nn.ReLU:
            return "relu", 0.0, rng
        if type(a) is torch.nn.LeakyReLU and a.negative_slope > 0:
            return "leaky", float(a.negative_slope), rng
        if type(a) is torch.nn.CELU and a.alpha == 1.0:
            return "celu", 0.0, rng
        if type(a) is RReLU and 0 < a.lower <= a.upper:
            if self.training:
                return "rrelu", 0.0, (float(a.lower), float(a.upper), drop_p)
            return "leaky", (float(a.lower) + float(a.upper)) / 2, None      # eval: the fixed mean slope
        return None                                  # PReLU (learnable slope): stays on torch

    def _next_node(self, n_rows, fa):
        """``ops.next_node_spec`` of this block's conv when the rows the GRU tail is about to write reach that conv unchanged in the next
        application: no norm, and a dropout slot that is empty, inactive, or the training-mode Dropout whose mask the tail draws."""
        conv = self.conv.conv if isinstance(self.conv, _TripletMessage) else None
        if conv is None or not isinstance(self.norm, _None):
            return None
        d = self.dropout
        live = type(d) is Dropout and self.training and 0 < d.p
        if not (isinstance(d, _None) or type(d) is Dropout) or (live and not (d.p < 1 and fa[2] is not None and fa[2][2] == float(d.p))):
            return None
        return ops.next_node_spec(conv, n_rows)

    def forward(self, x, edge_index, edge_attr, h=None, batch=None):
        identity = x
        seeded = h is None
        if seeded:
            h = x.unsqueeze(0)                       # layer.py:254 (pre-norm x seeds the GRU state)
        if (self.res is not False and batch is not None and isinstance(self.norm, (_PairNorm, _LayerNorm)) and x.is_cuda
                and torch.is_grad_enabled() and x.requires_grad):
            # x feeds the norm and the skip connection: the norm node hands x back as `identity`, so both gradient paths meet in its
            # backward kernel (one add launch per application less)
            # ... and with it the training-mode Dropout that follows (layer.py:255-256: run.py's PairNorm + Dropout(0.2)): one launch
            drop_p = self._norm_drop_p(x, batch)
            x, identity = self.norm(x, batch, with_identity=True, drop_p=drop_p) if drop_p > 0 else self.norm(x, batch, with_identity=True)
            if seeded:
                h = identity.unsqueeze(0)            # ... and the GRU state of the first application is that same tensor: one gradient
        else:
            drop_p = self._norm_drop_p(x, batch)
            x = self.norm(x, batch, drop_p=drop_p) if drop_p > 0 else self.norm(x, batch)
        if drop_p == 0:
            x = _apply_dropout(self.dropout, x)
        fa = self._fusable_act()
        if self.gru is None and isinstance(self.conv, (_GCNConv, _GATConv)) and fa is not None:
            # no GRU (layer.py:248): conv bias + residual + activation as one launch per direction
            c = self.conv.conv
            if (self.res is not False and identity is x and hasattr(c, "forward_with_identity") and x.is_cuda and torch.is_grad_enabled()
                    and x.requires_grad and ops.SKIP_THROUGH_CONV):
                # x feeds the conv and the skip connection: the x @ W node hands x back as `identity`, both gradient paths meet in the
                # epilogue of its d_x product (one add launch per application less)
                y, identity = c.forward_with_identity(x, edge_index)
            else:
                y = c(x, edge_index, add_bias=False)
            return ops.bias_res_act(y, c.bias, None if self.res is False else identity, fa[0], fa[1], rng=fa[2]), h
        if (self.res is not False and identity is x and hasattr(self.conv, "forward_with_identity") and x.is_cuda
                and torch.is_grad_enabled() and x.requires_grad and ops.SKIP_THROUGH_CONV):
            # x feeds the conv and the skip connection (no norm, no dropout between them): the conv node hands x back as `identity`,
            # both gradient paths meet in the epilogue of its d_x product (one add launch per application less)
            first = h.dim() == 3 and h.size(0) == 1 and h.data_ptr() == x.data_ptr() and h.shape[1:] == x.shape
            x, identity = self.conv.forward_with_identity(x, edge_index, edge_attr)
            if first:
                h = identity.unsqueeze(0)            # (layer.py:254: the GRU state is seeded with the same tensor)
        else:
            x = self.conv(x, edge_index, edge_attr)  # layer.py:259
        if self.gru is not None:
            g = self.gru
            if fa is not None and (fa[2] is None or ops.gru_rng_supported(x.size(1), g.weight_ih_l0, g.bias_ih_l0, g.bias_hh_l0)):
                # CELU (layer.py:261) + GRU step + residual + activation: one autograd node — which, when this block is applied again
                # (model.py:53-54) and nothing but the fused Dropout stands between its output and its conv, also writes the node
                # product of that next application
                x, hn = ops.gru_tail(x, h.squeeze(0), None if self.res is False else identity, g.weight_ih_l0, g.weight_hh_l0,
                                     g.bias_ih_l0, g.bias_hh_l0, act=fa[0], slope=fa[1], celu_in=True, rng=fa[2],
                                     node=self._next_node(x.size(0), fa))
                return x, hn.unsqueeze(0)
            if fa is not None and fa[0] != "rrelu":  # odd widths in training mode: fused tail without the dropped twin
                x, hn = ops.gru_tail(x, h.squeeze(0), None if self.res is False else identity, g.weight_ih_l0, g.weight_hh_l0,
                                     g.bias_ih_l0, g.bias_hh_l0, act=fa[0], slope=fa[1], celu_in=True)
                return x, hn.unsqueeze(0)
            x = torch.celu(x)                        # layer.py:261
            x = self._gru_step(x, h.squeeze(0))
            h = x.unsqueeze(0)
        x = x if self.res is False else x + identity
        x = _apply_act(self.act, x)
        return x, h


def dot_and_global_pool5(mol_out, pro_out, mol_batch, pro_batch):   # src_1gp/layer.py:270-283
    """[max, mean, median, min, std] of the ligand x residue score matrix of every pair: one HIP launch per direction (the
    reference loops over pairs in Python with a matmul, five reductions and .item() syncs each)."""
    msp = ops.segment_ptr(mol_batch)
    psp = ops.segment_ptr(pro_batch, msp.B)
    Cp = _ceil4(mol_out.size(1))
    if Cp > 128:
        raise GlamHipError(f"dot_and_global_pool5: width {mol_out.size(1)} > 128 is outside the compiled kernel table")
    return ops.pair_pool5(ops.pad_cols(mol_out, Cp), ops.pad_cols(pro_out, Cp), msp, psp)


def dot_and_global_pool2(mol_out, pro_out, mol_batch, pro_batch, with_identity=False):   # src_2gi_dti_scr/layer.py:270-283
    """[max, mean] of the ligand x residue score matrix of every pair: one HIP launch (no per-pair loop / syncs).  ``with_identity``:
    ``(out, mol_out, pro_out)`` — the two matrices handed back through the fusion node for their next use (see ops.pair_pool)."""
    msp = ops.segment_ptr(mol_batch)
    psp = ops.segment_ptr(pro_batch, msp.B)
    return ops.pair_pool(mol_out, pro_out, msp, psp, with_identity)
