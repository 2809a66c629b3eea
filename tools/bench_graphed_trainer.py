"""Epoch time of the reference-style training loop (fixed batch order, Adam, MSE) issued eagerly, through
glam_amd.graphs.GraphedTrainStep (one hipGraph per cached batch) and through GraphedTrainStep.run (16 consecutive steps per graph launch)."""
import sys, os, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from glam_amd import model, optim
from glam_amd.data import DataLoader, synth_molecule
from glam_amd.graphs import GraphedTrainStep

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
mols = [synth_molecule(rng) for _ in range(1128)]            # ESOL-sized dataset
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32            # run.py:40 default batch size
torch.manual_seed(0)
net0 = model.Architecture(mol_block="_TripletMessage", message_steps=3, mol_readout="GlobalPool5", graph_norm="_None",
                          graph_do="_None()", end_do="_None()", pre_act="ReLU", graph_act="ReLU", flat_act="ReLU").to(dev)
loss_fn = lambda out, b: torch.nn.functional.mse_loss(out.view(-1), b.y.view(-1))
for graphed in (False, True, "run"):
    net = copy.deepcopy(net0)
    opt = (optim.Adam(net.parameters(), lr=1e-3) if os.environ.get("GLAM_ADAM", "glam") == "glam"
           else torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True, fused=True))
    loader = DataLoader(mols, batch_size=B, device=dev)
    stepper = GraphedTrainStep(net, opt, loss_fn)
    times = []
    for epoch in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if graphed == "run":
            stepper.run(loader, steps_per_graph=16)
        for b in (loader if graphed != "run" else ()):
            if graphed:
                stepper(b)
            else:
                opt.zero_grad(set_to_none=True)
                loss_fn(net(b), b).backward()
                opt.step()
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    print(f"batch={B} {'16/launch' if graphed == 'run' else 'graphed  ' if graphed else 'eager    '}: epoch times (s) " + " ".join(f"{t:.3f}" for t in times)
          + f"   steady state {1128 / times[-1]:.0f} molecules/s", flush=True)
