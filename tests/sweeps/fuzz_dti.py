"""Randomised two-tower parity sweep (ArchitectureDTI on the HIP path vs the oracle): tower blocks, norms, pair counts and
protein-size mixes on both sides of the block-per-graph / wave-per-graph dispatch.  usage: python tests/sweeps/fuzz_dti.py [n] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from glam_amd import model
from glam_amd.data import synth_batch, synth_protein_batch
import oracle.glam_oracle as O

dev = torch.device("cuda")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
blocks = ["_TripletMessage", "_NNConv", "_TripletMessageLight", "_GCNConv", "_GATConv"]
bad = 0
for case in range(n_cases):
    lo = int(rng.choice([5, 20, 60]))
    cfg = dict(mol=str(rng.choice(blocks)), pro=str(rng.choice(blocks)), norm=str(rng.choice(["_None", "_PairNorm", "_LayerNorm"])),
               act=str(rng.choice(["ReLU", "CELU", "LeakyReLU"])), steps=int(rng.integers(1, 4)), P=int(rng.integers(1, 7)),
               n_min=lo, n_max=lo + int(rng.choice([0, 30, 150])), alpha=int(rng.choice([2, 3, 4])))
    try:
        torch.manual_seed(500 + case)
        mb = synth_batch(cfg["P"], seed=case)
        pb = synth_protein_batch(cfg["P"], seed=1000 + case, n_min=cfg["n_min"], n_max=cfg["n_max"])
        net = model.ArchitectureDTI(hid_dim_alpha=cfg["alpha"], mol_block=cfg["mol"], pro_block=cfg["pro"], e_dim=64, message_steps=cfg["steps"],
                                    graph_norm=cfg["norm"], pre_act=cfg["act"], graph_act=cfg["act"], flat_act=cfg["act"], end_act=cfg["act"],
                                    graph_do="_None()", end_do="_None()").eval()
        sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
        ref = O.architecture_dti(sd, mb, pb, cfg["P"], message_steps=cfg["steps"], mol_block=cfg["mol"], pro_block=cfg["pro"], graph_norm=cfg["norm"],
                                 pre_act=cfg["act"], graph_act=cfg["act"], flat_act=cfg["act"], end_act=cfg["act"])
        cot = torch.randn(ref.shape)
        names = [n for n, _ in net.named_parameters()]
        g_ref = torch.autograd.grad((ref * cot).sum(), [sd[n] for n in names], allow_unused=True)
        net = net.to(dev)
        out = net(mb.to(dev), pb.to(dev))
        err = (out.cpu() - ref).abs().max().item()
        assert err <= 5e-5 * max(1.0, ref.abs().max().item()), f"out {err:.2e}"
        gs = torch.autograd.grad((out * cot.to(dev)).sum(), [p for _, p in net.named_parameters()], allow_unused=True)
        for n, a, r in zip(names, gs, g_ref):
            if r is None:
                continue
            e = (a.cpu() - r).abs().max().item()
            lim = 3e-4 * max(1.0, r.abs().max().item())
            assert e <= lim, f"grad {n}: {e:.2e} > {lim:.2e}"
        print("ok  ", cfg, flush=True)
    except Exception as e:   # noqa: BLE001
        bad += 1
        print("FAIL", cfg, "->", type(e).__name__, str(e)[:200], flush=True)
print(f"{n_cases - bad}/{n_cases} cases passed")
sys.exit(1 if bad else 0)
