"""Stand-in for ``torch_geometric.nn`` (test infrastructure; see package docstring).

Pools / norms / readouts restate the PyG 1.7.2 semantics listed in SURVEY.md §8c.
"""
import torch
from torch.nn import Parameter

from .conv import MessagePassing, NNConv, GCNConv, GATConv  # noqa: F401
from ..utils import scatter, softmax, degree, to_dense_batch


def _num_graphs(batch):
    return int(batch.max()) + 1


def global_add_pool(x, batch, size=None):
    return scatter(x, batch, 0, size or _num_graphs(batch), "sum")


def global_mean_pool(x, batch, size=None):
    return scatter(x, batch, 0, size or _num_graphs(batch), "mean")


def global_max_pool(x, batch, size=None):
    return scatter(x, batch, 0, size or _num_graphs(batch), "max")


def global_sort_pool(x, batch, k):
    """Per graph: sort nodes by LAST channel (descending), keep the first ``k`` full
    rows, zero-pad short graphs, flatten to ``[B, k*D]``."""
    fill_value = x.min().item() - 1
    dense, _ = to_dense_batch(x, batch, fill_value)
    B, nmax, D = dense.shape
    _, perm = dense[:, :, -1].sort(dim=-1, descending=True)
    arange = torch.arange(B, dtype=torch.long, device=x.device) * nmax
    perm = perm + arange.view(-1, 1)
    dense = dense.view(B * nmax, D)[perm.view(-1)].view(B, nmax, D)
    if nmax >= k:
        dense = dense[:, :k].contiguous()
    else:
        pad = dense.new_full((B, k - nmax, D), fill_value)
        dense = torch.cat([dense, pad], dim=1)
    dense[dense == fill_value] = 0
    return dense.view(B, k * D)


class GlobalAttention(torch.nn.Module):
    def __init__(self, gate_nn, nn=None):
        super().__init__()
        self.gate_nn, self.nn = gate_nn, nn

    def forward(self, x, batch, size=None):
        x = x.unsqueeze(-1) if x.dim() == 1 else x
        size = int(batch[-1]) + 1 if size is None else size
        gate = self.gate_nn(x).view(-1, 1)
        x = self.nn(x) if self.nn is not None else x
        gate = softmax(gate, batch, num_nodes=size)
        return scatter(gate * x, batch, 0, size, "sum")


class Set2Set(torch.nn.Module):
    def __init__(self, in_channels, processing_steps, num_layers=1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, 2 * in_channels
        self.processing_steps, self.num_layers = processing_steps, num_layers
        self.lstm = torch.nn.LSTM(self.out_channels, self.in_channels, num_layers)

    def forward(self, x, batch):
        B = _num_graphs(batch)
        h = (x.new_zeros((self.num_layers, B, self.in_channels)),
             x.new_zeros((self.num_layers, B, self.in_channels)))
        q_star = x.new_zeros(B, self.out_channels)
        for _ in range(self.processing_steps):
            q, h = self.lstm(q_star.unsqueeze(0), h)
            q = q.view(B, self.in_channels)
            e = (x * q[batch]).sum(dim=-1, keepdim=True)
            a = softmax(e, batch, num_nodes=B)
            r = scatter(a * x, batch, 0, B, "sum")
            q_star = torch.cat([q, r], dim=-1)
        return q_star


class BatchNorm(torch.nn.Module):
    def __init__(self, in_channels, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.module = torch.nn.BatchNorm1d(in_channels, eps, momentum, affine, track_running_stats)

    def forward(self, x):
        return self.module(x)


class LayerNorm(torch.nn.Module):
    """PyG graph LayerNorm: statistics over ALL nodes x channels of a graph."""

    def __init__(self, in_channels, eps=1e-5, affine=True):
        super().__init__()
        self.eps = eps
        self.weight = Parameter(torch.ones(in_channels)) if affine else None
        self.bias = Parameter(torch.zeros(in_channels)) if affine else None

    def forward(self, x, batch=None):
        if batch is None:
            x = x - x.mean()
            out = x / (x.std(unbiased=False) + self.eps)
        else:
            B = _num_graphs(batch)
            norm = degree(batch, B, dtype=x.dtype).clamp_(min=1)
            norm = norm.mul_(x.size(-1)).view(-1, 1)
            mean = scatter(x, batch, 0, B, "sum").sum(dim=-1, keepdim=True) / norm
            x = x - mean[batch]
            var = scatter(x * x, batch, 0, B, "sum").sum(dim=-1, keepdim=True) / norm
            out = x / (var + self.eps).sqrt()[batch]
        if self.weight is not None:
            out = out * self.weight + self.bias
        return out


class PairNorm(torch.nn.Module):
    def __init__(self, scale=1.0, scale_individually=False, eps=1e-5):
        super().__init__()
        assert not scale_individually
        self.scale, self.eps = scale, eps

    def forward(self, x, batch=None):
        if batch is None:
            x = x - x.mean(dim=0, keepdim=True)
            return self.scale * x / (self.eps + x.pow(2).sum(-1).mean()).sqrt()
        mean = scatter(x, batch, 0, _num_graphs(batch), "mean")
        x = x - mean[batch]
        sq = scatter(x.pow(2).sum(-1, keepdim=True), batch, 0, _num_graphs(batch), "mean")
        return self.scale * x / (self.eps + sq[batch]).sqrt()


class GraphSizeNorm(torch.nn.Module):
    def forward(self, x, batch=None):
        if batch is None:
            batch = torch.zeros(x.size(0), dtype=torch.long, device=x.device)
        inv_sqrt_deg = degree(batch, dtype=x.dtype).pow(-0.5)
        return x * inv_sqrt_deg[batch].view(-1, 1)


class InstanceNorm(torch.nn.Module):  # imported by the reference, never instantiated
    def __init__(self, *a, **k):
        raise NotImplementedError("InstanceNorm is not used on the GLAM path")


class MessageNorm(torch.nn.Module):  # imported by the reference, never instantiated
    def __init__(self, *a, **k):
        raise NotImplementedError("MessageNorm is not used on the GLAM path")
