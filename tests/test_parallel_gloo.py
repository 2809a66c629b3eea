"""CPU, world_size 2 over gloo: the data-parallel path (shard by graph, one flat gradient bucket, one
all-reduce per step) reproduces the single-process gradients of the concatenated batch.

The HIP layers need a GPU, so the replicated module here is a small pure-torch graph model with the
same structure (per-node linear -> per-graph mean readout -> head): what is under test is
glam_amd.parallel, which is model-agnostic."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from glam_amd.data import synth_batch
from glam_amd.parallel import (DataParallelStep, FlatGradBucket, balanced_graph_ranges, masked_loss_weight,
                               shard_batch)


class TinyGraphNet(torch.nn.Module):
    def __init__(self, tasks=1):
        super().__init__()
        self.lin = torch.nn.Linear(15, 8)
        self.head = torch.nn.Linear(8, tasks)

    def forward(self, b):
        h = torch.relu(self.lin(b.x))
        B = b.num_graphs
        s = torch.zeros(B, 8).index_add_(0, b.batch, h)
        cnt = torch.zeros(B).index_add_(0, b.batch, torch.ones(b.batch.numel())).clamp_(min=1)
        return self.head(s / cnt.view(-1, 1))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(1234 + rank)          # different seeds: broadcast_parameters must align the replicas
        tasks = 1 if mode == "mse" else 3
        net = TinyGraphNet(tasks)
        full = synth_batch(24, seed=5, n_tasks=tasks, task="regression" if mode == "mse" else "classification")
        shard = shard_batch(full, rank, world)
        Btot = 24

        if mode == "mse":
            def loss_fn(out, b):
                # sum over the shard / global graph count == this rank's share of the global mean
                return ((out.view(-1) - b.y.view(-1)) ** 2).sum() / Btot, torch.tensor(1.0)
        else:
            def loss_fn(out, b):
                mask = b.y >= 0
                n_valid = mask.sum()
                local = torch.nn.functional.binary_cross_entropy_with_logits(out[mask], b.y[mask])
                return local, masked_loss_weight(n_valid)

        step = DataParallelStep(net, loss_fn)
        step(shard)
        # plain Python payloads: tensors through an mp.Queue hand over file descriptors the exiting worker may close
        q.put((rank, step.bucket.flat.tolist(), [p.detach().reshape(-1).tolist() for p in net.parameters()]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["mse", "masked_bce"])
def test_dp_gradients_match_single_process(mode):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, g0, p0), (_, g1, p1) = res
    assert g0 == g1, "all-reduced buckets must be identical on every rank"
    assert p0 == p1, "replicas must start from rank 0's parameters"
    g0 = torch.tensor(g0)
    # single-process reference on the whole batch with rank 0's parameters
    tasks = 1 if mode == "mse" else 3
    net = TinyGraphNet(tasks)
    with torch.no_grad():
        for p, v in zip(net.parameters(), p0):
            p.copy_(torch.tensor(v).view_as(p))
    full = synth_batch(24, seed=5, n_tasks=tasks, task="regression" if mode == "mse" else "classification")
    out = net(full)
    if mode == "mse":
        loss = torch.nn.functional.mse_loss(out.view(-1), full.y.view(-1))
    else:
        mask = full.y >= 0
        loss = torch.nn.functional.binary_cross_entropy_with_logits(out[mask], full.y[mask])
    loss.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    assert torch.allclose(g0, ref, rtol=1e-5, atol=1e-6), (g0 - ref).abs().max()


def test_shard_batch_partitions_graphs_exactly():
    full = synth_batch(37, seed=2)
    for world in (1, 2, 3, 8):
        shards = [shard_batch(full, r, world) for r in range(world)]
        assert sum(s.num_graphs for s in shards) == 37
        assert sum(s.x.size(0) for s in shards) == full.x.size(0)
        assert sum(s.edge_index.size(1) for s in shards) == full.edge_index.size(1)
        off = 0
        for s in shards:
            n = s.x.size(0)
            assert torch.equal(s.x, full.x[off:off + n])
            if n:
                assert int(s.edge_index.min()) >= 0 and int(s.edge_index.max()) < n
                assert int(s.batch.min()) == 0 and int(s.batch.max()) == s.num_graphs - 1
            off += n
        counts = [s.x.size(0) for s in shards]
        assert max(counts) - min(counts) <= 2 * 28, "node-balanced contiguous ranges"


def test_balanced_ranges_cover_and_handle_skew():
    sizes = [800, 10, 10, 10, 700, 15, 600, 5]
    r = balanced_graph_ranges(sizes, 4)
    assert r[0][0] == 0 and r[-1][1] == len(sizes) and all(r[i][1] == r[i + 1][0] for i in range(3))
    assert balanced_graph_ranges([5], 4)[-1][1] == 1


def test_flat_bucket_views_alias_parameters():
    net = TinyGraphNet()
    bucket = FlatGradBucket(net.parameters())
    assert bucket.flat.numel() == sum(p.numel() for p in net.parameters())
    b = synth_batch(4, seed=0)
    net(b).sum().backward()
    assert bucket.flat.abs().sum() > 0
    got = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    assert torch.equal(got, bucket.flat) and all(p.grad.data_ptr() >= bucket.flat.data_ptr() for p in net.parameters())
    bucket.zero()
    assert all(float(p.grad.abs().sum()) == 0 for p in net.parameters())
