"""Eager CPU-side cost of one training step at batch 32, split into forward / backward / optimizer (A/B aid)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from glam_amd import model
from glam_amd.data import DataLoader, synth_molecule
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
mols = [synth_molecule(rng) for _ in range(1128)]
torch.manual_seed(0)
net = model.Architecture(mol_block="_TripletMessage", message_steps=3, mol_readout="GlobalPool5", graph_norm="_None",
                          graph_do="_None()", end_do="_None()", pre_act="ReLU", graph_act="ReLU", flat_act="ReLU").to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True, fused=True)
batches = list(DataLoader(mols, batch_size=32, device=dev))
loss_fn = lambda out, b: torch.nn.functional.mse_loss(out.view(-1), b.y.view(-1))
tf = tb = to = 0.0
for ep in range(4):
    for b in batches:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss = loss_fn(net(b), b)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        opt.step()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        if ep >= 2:
            tf += t1 - t0; tb += t2 - t1; to += t3 - t2
n = 2 * len(batches)
print(f"{os.getcwd()}: forward {tf / n * 1e3:.3f} ms  backward {tb / n * 1e3:.3f} ms  optimizer {to / n * 1e3:.3f} ms")
