"""One-process-per-GPU data parallelism for the message-passing path (new capability: the
reference has no ``torch.distributed`` use at all — SURVEY.md §2 / §8e).

A collated batch is a disjoint union of graphs and no edge crosses graphs, so the path shards by
*graph* with no data-path collective: every rank takes a contiguous range of graphs (balanced by
node count, proteins vary 10x), stages its own CSR and runs the same kernels.  The only exchange is
one all-reduce of ONE flat fp32 gradient bucket per step (≈0.36 M parameters ≈ 1.4 MB for the
default model): over xGMI that message is latency-bound, so it is issued as a single RCCL call
(backend ``"nccl"`` is RCCL on ROCm) rather than per-parameter or ring-bucketed pieces.

Works unchanged with the ``gloo`` backend on CPU tensors (used by the CPU test-suite).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .data import Batch


def balanced_graph_ranges(nodes_per_graph, world_size):
    """Contiguous graph ranges ``[(g0, g1), ...]`` whose node counts are as even as a prefix split
    allows (greedy on the node-count prefix sum).  Every rank gets a (possibly empty) range."""
    sizes = np.asarray(nodes_per_graph, dtype=np.int64)
    csum = np.concatenate([[0], np.cumsum(sizes)])
    total = int(csum[-1])
    cuts = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        g = int(np.searchsorted(csum, target, side="left"))
        if g > 0 and abs(csum[g - 1] - target) <= abs(csum[min(g, len(sizes))] - target):
            g -= 1
        cuts.append(min(max(g, cuts[-1]), len(sizes)))
    cuts.append(len(sizes))
    return [(cuts[i], cuts[i + 1]) for i in range(world_size)]


def shard_batch(batch, rank, world_size):
    """The sub-batch of graphs owned by ``rank`` (host side; call before ``.to(device)``).
    Node ids in ``edge_index`` and graph ids in ``batch`` are re-based to the shard."""
    B = int(batch.batch[-1]) + 1 if batch.batch.numel() else 0
    counts = torch.bincount(batch.batch, minlength=B)
    g0, g1 = balanced_graph_ranges(counts.numpy(), world_size)[rank]
    ptr = torch.cat([counts.new_zeros(1), counts.cumsum(0)])
    n0, n1 = int(ptr[g0]), int(ptr[g1])
    src = batch.edge_index[0]
    emask = (src >= n0) & (src < n1)
    out = Batch(x=batch.x[n0:n1], edge_index=batch.edge_index[:, emask] - n0,
                edge_attr=None if batch.edge_attr is None else batch.edge_attr[emask],
                y=None if batch.y is None else batch.y[g0:g1], batch=batch.batch[n0:n1] - g0)
    out.num_graphs = g1 - g0
    out.graph_range = (g0, g1)
    return out


class FlatGradBucket:
    """All parameter gradients as views into ONE contiguous fp32 buffer, so a step needs exactly one
    all-reduce.  Autograd accumulates into the views; ``zero()`` clears the bucket in one memset."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(n, dtype=p0.dtype, device=p0.device)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce(self, scale=None, group=None):
        """Sum over ranks, then multiply by ``scale`` (``1/world`` for a mean-reduced loss)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        if scale is not None:
            self.flat.mul_(scale)


def flat_view(grads):
    """The single contiguous tensor that ``grads`` are consecutive views of, or ``None``.  The fused layer returns its
    parameter gradients this way (DDP's gradient-as-bucket-view, without registering hooks): the bucket all-reduce
    then needs no gather copy, and the reduced values are visible through the individual gradients."""
    if not grads:
        return None
    g0 = grads[0]
    base = g0._base if g0._base is not None else g0
    off = g0.storage_offset()
    for g in grads:
        gb = g._base if g._base is not None else g
        if gb is not base or not g.is_contiguous() or g.storage_offset() != off or g.dtype != g0.dtype:
            return None
        off += g.numel()
    total = off - g0.storage_offset()
    return torch.as_strided(base, (total,), (1,), g0.storage_offset())


def broadcast_parameters(module, src=0):
    """Replicas start identical (same seed gives the same init; this makes it unconditional)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)


def masked_loss_weight(n_valid_local):
    """For losses averaged over *valid labels only* (masked BCE, ``src_1gp/trainer.py:244-245``):
    the factor that turns the sum over ranks of local-mean gradients into the gradient of the
    global mean, ``n_valid_local / n_valid_global`` (one scalar all-reduce)."""
    t = n_valid_local.clone().to(torch.float32).reshape(1)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return n_valid_local.to(torch.float32) / t.clamp(min=1.0)


class DataParallelStep:
    """``zero -> forward -> loss -> backward -> one all-reduce`` for a replicated module.

    ``loss_fn(output, shard) -> (loss, weight)``: ``weight`` scales this rank's gradient before the
    sum over ranks (``1/world`` for equal shards of a mean loss; see ``masked_loss_weight``)."""

    def __init__(self, module, loss_fn):
        self.module, self.loss_fn = module, loss_fn
        broadcast_parameters(module)
        self.bucket = FlatGradBucket(module.parameters())

    def __call__(self, *shard_inputs):
        self.bucket.zero()
        out = self.module(*shard_inputs)
        loss, weight = self.loss_fn(out, *shard_inputs)
        (loss * weight).backward()
        self.bucket.all_reduce()
        return loss.detach()
