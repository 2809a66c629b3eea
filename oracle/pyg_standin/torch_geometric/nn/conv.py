"""Stand-in for ``torch_geometric.nn.conv`` (test infrastructure; see package docstring).

``MessagePassing`` restates SURVEY.md Appendix B: flow ``source_to_target`` =>
``_j`` arguments are gathered with ``edge_index[0]``, ``_i`` with ``edge_index[1]``,
the aggregate scatters at ``edge_index[1]`` with ``dim_size = N``.
"""
import inspect
import math

import torch
from torch.nn import Parameter, Linear
import torch.nn.functional as F

from ..utils import scatter, softmax, add_remaining_self_loops, remove_self_loops, add_self_loops


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2):
        super().__init__()
        assert flow == "source_to_target"
        self.aggr = aggr
        self.flow = flow
        self.node_dim = node_dim
        self._msg_params = [p for p in inspect.signature(self.message).parameters]

    def propagate(self, edge_index, size=None, **kwargs):
        dim = self.node_dim
        n_src = n_dst = None
        args = {}
        for name in self._msg_params:
            if name.endswith("_j") or name.endswith("_i"):
                base, which = name[:-2], name[-1]
                if base == "edge_index":
                    args[name] = edge_index[0] if which == "j" else edge_index[1]
                    continue
                if base == "size":
                    continue
                data = kwargs[base]
                if isinstance(data, (tuple, list)):
                    data = data[0] if which == "j" else data[1]
                if which == "j":
                    n_src = data.size(dim)
                    args[name] = data.index_select(dim, edge_index[0])
                else:
                    n_dst = data.size(dim)
                    args[name] = data.index_select(dim, edge_index[1])
            elif name in ("index",):
                args[name] = edge_index[1]
            elif name in ("ptr",):
                args[name] = None
            else:
                args[name] = kwargs.get(name)
        if size is not None and size[1] is not None:
            n_dst = size[1]
        if n_dst is None:
            n_dst = n_src
        if n_src is None:
            n_src = n_dst
        if "size_i" in self._msg_params:
            args["size_i"] = n_dst
        if "size_j" in self._msg_params:
            args["size_j"] = n_src
        msg = self.message(**args)
        out = scatter(msg, edge_index[1], 0 if dim in (0, -msg.dim()) else dim, n_dst, self.aggr)
        return self.update(out)

    def message(self, x_j):
        return x_j

    def update(self, aggr_out):
        return aggr_out


def _glorot(t):
    stdv = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-stdv, stdv)


def _uniform(size, t):
    bound = 1.0 / math.sqrt(size)
    with torch.no_grad():
        t.uniform_(-bound, bound)


class NNConv(MessagePassing):
    """``x_i' = x_i @ root + aggr_j ( x_j @ nn(e_ij).view(in, out) ) + bias`` (PyG 1.7.2)."""

    def __init__(self, in_channels, out_channels, nn, aggr="add", root_weight=True, bias=True):
        super().__init__(aggr=aggr, node_dim=0)
        self.in_channels, self.out_channels, self.nn = in_channels, out_channels, nn
        self.root = Parameter(torch.Tensor(in_channels, out_channels)) if root_weight else None
        self.bias = Parameter(torch.Tensor(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        for m in self.nn.modules():
            if m is not self.nn and hasattr(m, "reset_parameters"):
                m.reset_parameters()
        if self.root is not None:
            _uniform(self.root.size(0), self.root)
        if self.bias is not None:
            torch.nn.init.zeros_(self.bias)

    def forward(self, x, edge_index, edge_attr, size=None):
        out = self.propagate(edge_index, x=x, edge_attr=edge_attr, size=size)
        if self.root is not None:
            out = out + torch.matmul(x, self.root)
        if self.bias is not None:
            out = out + self.bias
        return out

    def message(self, x_j, edge_attr):
        weight = self.nn(edge_attr).view(-1, self.in_channels, self.out_channels)
        return torch.matmul(x_j.unsqueeze(1), weight).squeeze(1)


class GCNConv(MessagePassing):
    """``D^-1/2 (A + I) D^-1/2 X W + b`` (PyG 1.7.2 defaults)."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__(aggr="add", node_dim=0)
        self.weight = Parameter(torch.Tensor(in_channels, out_channels))
        self.bias = Parameter(torch.Tensor(out_channels)) if bias else None
        _glorot(self.weight)
        if self.bias is not None:
            torch.nn.init.zeros_(self.bias)

    def forward(self, x, edge_index, edge_weight=None):
        n = x.size(0)
        if edge_weight is None:
            edge_weight = x.new_ones(edge_index.size(1))
        edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight, 1.0, n)
        row, col = edge_index
        deg = scatter(edge_weight, col, 0, n, "sum")
        dis = deg.pow(-0.5)
        dis = dis.masked_fill(dis == float("inf"), 0)
        norm = dis[row] * edge_weight * dis[col]
        x = torch.matmul(x, self.weight)
        out = self.propagate(edge_index, x=x, edge_weight=norm)
        if self.bias is not None:
            out = out + self.bias
        return out

    def message(self, x_j, edge_weight):
        return edge_weight.view(-1, 1) * x_j


class GATConv(MessagePassing):
    """Single-``Linear`` GAT (PyG 1.7.2, ``heads=1, concat=True, add_self_loops=True``)."""

    def __init__(self, in_channels, out_channels, heads=1, negative_slope=0.2, bias=True):
        super().__init__(aggr="add", node_dim=0)
        self.heads, self.out_channels, self.negative_slope = heads, out_channels, negative_slope
        self.lin_l = Linear(in_channels, heads * out_channels, bias=False)
        self.lin_r = self.lin_l
        self.att_l = Parameter(torch.Tensor(1, heads, out_channels))
        self.att_r = Parameter(torch.Tensor(1, heads, out_channels))
        self.bias = Parameter(torch.Tensor(heads * out_channels)) if bias else None
        _glorot(self.lin_l.weight)
        _glorot(self.att_l)
        _glorot(self.att_r)
        if self.bias is not None:
            torch.nn.init.zeros_(self.bias)

    def forward(self, x, edge_index):
        H, C = self.heads, self.out_channels
        xl = self.lin_l(x).view(-1, H, C)
        al = (xl * self.att_l).sum(-1)
        ar = (xl * self.att_r).sum(-1)
        edge_index, _ = remove_self_loops(edge_index)
        edge_index, _ = add_self_loops(edge_index, num_nodes=x.size(0))
        out = self.propagate(edge_index, x=xl, alpha=(al, ar))
        out = out.view(-1, H * C)
        if self.bias is not None:
            out = out + self.bias
        return out

    def message(self, x_j, alpha_j, alpha_i, edge_index_i, size_i):
        alpha = F.leaky_relu(alpha_j + alpha_i, self.negative_slope)
        alpha = softmax(alpha, edge_index_i, None, size_i)
        return x_j * alpha.unsqueeze(-1)
