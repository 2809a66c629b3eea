"""GraphedTrainStep vs the eager loop on random model configurations: identical parameter trajectories (bit for bit).
usage: python tools/fuzz_graphed.py [n_cases] [seed]"""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from glam_amd import model, optim
from glam_amd.data import synth_batch
from glam_amd.graphs import GraphedTrainStep

dev = torch.device("cuda")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    cfg = dict(alpha=int(rng.choice([1, 2, 3, 4, 6])), block=str(rng.choice(["_TripletMessage", "_NNConv", "_TripletMessageLight", "_GCNConv", "_GATConv"])),
               readout=str(rng.choice(["GlobalPool5", "GlobalLAPool", "Set2Set"])), norm=str(rng.choice(["_None", "_PairNorm", "_LayerNorm"])),
               act=str(rng.choice(["ReLU", "CELU", "LeakyReLU"])), steps=int(rng.integers(1, 4)))
    try:
        torch.manual_seed(case)
        net0 = model.Architecture(hid_dim_alpha=cfg["alpha"], e_dim=64, message_steps=cfg["steps"], mol_block=cfg["block"], mol_readout=cfg["readout"],
                                  graph_norm=cfg["norm"], pre_act=cfg["act"], graph_act=cfg["act"], flat_act=cfg["act"], graph_do="_None()",
                                  end_do="_None()").to(dev)
        batches = [synth_batch(int(rng.integers(2, 40)), seed=100 * case + k).to(dev) for k in range(3)]
        loss_fn = lambda out, b: torch.nn.functional.mse_loss(out.view(-1), b.y.view(-1))
        finals = []
        for graphed in (False, True):
            net = copy.deepcopy(net0)
            opt = (optim.Adam(net.parameters(), lr=2.0 ** -10) if case % 2 else torch.optim.Adam(net.parameters(), lr=2.0 ** -10, capturable=True, fused=True))   # exactly representable: the graphed stepper keeps lr in an fp32 device tensor, the eager loop in a Python float
            stepper = GraphedTrainStep(net, opt, loss_fn)
            for epoch in range(4):
                for b in batches:
                    if graphed:
                        stepper(b)
                    else:
                        opt.zero_grad(set_to_none=True)
                        loss_fn(net(b), b).backward()
                        opt.step()
            torch.cuda.synchronize()
            finals.append([p.detach().clone() for p in net.parameters()])
        same = all(torch.equal(a, b) for a, b in zip(*finals))
        worst = max(float((a - b).abs().max()) for a, b in zip(*finals))
        assert same, f"trajectories differ (max {worst:.3e})"
        print("ok  ", cfg, flush=True)
    except Exception as e:   # noqa: BLE001
        bad += 1
        print("FAIL", cfg, "->", type(e).__name__, str(e)[:160], flush=True)
print(f"{n_cases - bad}/{n_cases} cases passed")
sys.exit(1 if bad else 0)
