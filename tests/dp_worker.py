"""One rank of tests/test_gpu_configs.py::test_config4_two_process_data_parallel_step_on_the_hip_model (not a test module)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    multi = torch.cuda.device_count() >= world
    dev = torch.device("cuda", rank if multi else 0)
    torch.cuda.set_device(dev)
    if multi:
        dist.init_process_group("nccl", device_id=dev)          # RCCL
    else:
        dist.init_process_group("gloo")
    from glam_amd import model
    from glam_amd.data import synth_batch
    from glam_amd.parallel import DataParallelStep, masked_loss_weight, shard_batch

    def masked_bce(o, y):
        m = y >= 0
        return torch.nn.functional.binary_cross_entropy_with_logits(o[m], y[m])

    torch.manual_seed(3)
    net = model.Architecture(mol_block="_TripletMessage", message_steps=3, mol_readout="GlobalPool5", e_dim=256, out_dim=12).eval()
    if rank != 0:                      # replicas must come out identical through broadcast_parameters, not through the seed
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    net = net.to(dev)
    full = synth_batch(96, seed=9, n_tasks=12, task="classification")
    step = DataParallelStep(net, lambda o, s: (masked_bce(o, s.y), masked_loss_weight((s.y >= 0).sum())))
    shard = shard_batch(full, rank, world).to(dev)
    step(shard)
    got = step.bucket.flat.clone()
    if rank == 0:
        net.zero_grad(set_to_none=True)
        fb = full.to(dev)
        loss = masked_bce(net(fb), fb.y)
        ref = torch.cat([g.reshape(-1) for g in torch.autograd.grad(loss, list(net.parameters()))])
        err = (got - ref).abs().max().item()
        scale = ref.abs().max().item()
        assert err <= 2e-5 * max(1.0, scale), f"two-rank gradient differs from the single-process one: {err:.3e} (scale {scale:.3g})"
        print(f"DP-OK backend={'nccl' if multi else 'gloo'} max|d|={err:.2e}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
