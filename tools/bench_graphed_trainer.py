"""Epoch time of the reference-style training loop (fixed batch order, Adam, MSE; src_1gp/trainer.py:286-304):
  eager      the loop as written, the model's graphed-callable route switched off (model.graphed_call = False)
  unchanged  the loop as written — ``model(batch)`` replays hipGraphs by itself (glam_amd.graphs.GraphedCallable), cached batch objects
  unch+fresh the loop as written on a FRESH device copy of every batch in every epoch (what trainer.py:294 hands the model: recognised
             by content fingerprint, one 8-byte read-back per step)
  graphed    glam_amd.graphs.GraphedTrainStep (the whole step incl. the optimizer in one hipGraph per cached batch: needs a trainer edit)
  16/launch  GraphedTrainStep.run (16 consecutive steps per graph launch)"""
import sys, os, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from glam_amd import model, optim
from glam_amd.data import Batch, DataLoader, synth_molecule
from glam_amd.graphs import GraphedTrainStep

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32            # run.py:40 default batch size
NMOL = max(1128, 8 * B)                                      # ESOL-sized dataset (at least eight batches)
mols = [synth_molecule(rng) for _ in range(NMOL)]
torch.manual_seed(0)
net0 = model.Architecture(mol_block="_TripletMessage", message_steps=3, mol_readout="GlobalPool5", graph_norm="_None",
                          graph_do="_None()", end_do="_None()", pre_act="ReLU", graph_act="ReLU", flat_act="ReLU").to(dev)
loss_fn = lambda out, b: torch.nn.functional.mse_loss(out.view(-1), b.y.view(-1))
def fresh(b):
    out = Batch(x=b.x.clone(), edge_index=b.edge_index.clone(), edge_attr=b.edge_attr.clone(), y=b.y.clone(), batch=b.batch.clone())
    out.num_graphs = b.num_graphs
    return out

for graphed in (False, "unchanged", "unch+fresh", True, "run"):
    net = copy.deepcopy(net0)
    net.graphed_call = graphed in ("unchanged", "unch+fresh")
    opt = (optim.Adam(net.parameters(), lr=1e-3) if os.environ.get("GLAM_ADAM", "glam") == "glam"
           else torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True, fused=True))
    loader = DataLoader(mols, batch_size=B, device=dev)
    stepper = GraphedTrainStep(net, opt, loss_fn)
    times = []
    for epoch in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if graphed == "run":
            stepper.run(loader, steps_per_graph=16)
        for b in (loader if graphed != "run" else ()):
            if graphed is True:
                stepper(b)
            else:
                if graphed == "unch+fresh":
                    b = fresh(b)
                opt.zero_grad()
                loss = loss_fn(net(b), b)
                loss.backward()
                opt.step()
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    steady = sorted(times[3:])[len(times[3:]) // 2]
    print(f"batch={B} {'16/launch ' if graphed == 'run' else 'graphed   ' if graphed is True else 'eager     ' if not graphed else graphed.ljust(10)}: epoch times (ms) " + " ".join(f"{t * 1e3:.2f}" for t in times)
          + f"   steady state {steady / len(loader) * 1e3:.4f} ms per step, {NMOL / steady:.0f} molecules/s", flush=True)
