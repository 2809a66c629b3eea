"""A SHUFFLING training loop end to end (new batch composition every step: collation, host-to-device copy, CSR staging and
its validation sync, eager step): molecules/s at B = 1024 and B = 32."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from glam_amd import model, optim
from glam_amd.data import DataLoader, synth_molecule
dev = torch.device("cuda")
rng = np.random.default_rng(0)
mols = [synth_molecule(rng) for _ in range(8192)]
torch.manual_seed(0)
net = model.Architecture(mol_block="_NNConv", graph_norm="_PairNorm", graph_do="_None()", end_do="_None()", pre_act="ReLU", graph_act="ReLU",
                         flat_act="ReLU").to(dev)
opt = (optim.Adam(net.parameters(), lr=1e-3) if os.environ.get("GLAM_ADAM", "glam") == "glam"
           else torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True, fused=True))
for B in (1024, 32):
    loader = DataLoader(mols if B == 1024 else mols[:2048], batch_size=B, shuffle=True, device=dev)
    times = []
    for epoch in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        t_col = 0.0
        it = iter(loader)
        while True:
            c0 = time.perf_counter()
            b = next(it, None)
            t_col += time.perf_counter() - c0
            if b is None:
                break
            opt.zero_grad(set_to_none=True)
            torch.nn.functional.mse_loss(net(b).view(-1), b.y.view(-1)).backward()
            opt.step()
        torch.cuda.synchronize(); times.append((time.perf_counter() - t0, t_col))
    n = len(loader.dataset)
    print(f"B={B}: epoch {times[-1][0] * 1e3:.1f} ms for {n} molecules = {n / times[-1][0]:.0f} molecules/s "
          f"(collate + copy: {times[-1][1] * 1e3:.1f} ms of it)", flush=True)
