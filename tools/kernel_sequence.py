"""Developer aid: the library's kernel launches of ONE full-model training step, in launch order (glam_prof_* labels and durations; ATen
launches are not listed: tools/bench_model.py --profile names those).  usage: kernel_sequence.py [batch] [preset]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glam_amd import _lib, model, ops, optim, loss as glam_loss
ops.USE_TORCH_EXT = False      # the route a captured step takes
from glam_amd.data import synth_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")
torch.manual_seed(0)
PRESET = sys.argv[2] if len(sys.argv) > 2 else "relu"
if PRESET == "model_default":      # Architecture()'s keyword defaults in training mode: RReLU x 3, Dropout(0.2) twice
    net = model.Architecture(mol_block="_TripletMessage", message_steps=3).to(dev).train()
else:
    net = model.Architecture(mol_block="_TripletMessage", message_steps=3, graph_do="_None()", end_do="_None()", pre_act="ReLU", graph_act="ReLU",
                             flat_act="ReLU").to(dev)
net.graphed_call = False      # (eager launches: every one carries its own events)
b = synth_batch(B, seed=0).to(dev)
y = b.y.view(-1)
opt = optim.Adam(net.parameters(), lr=1e-3)
ONE = torch.ones((), device=dev)

def body():
    opt.zero_grad(set_to_none=True)
    loss = glam_loss.mse_loss(net(b).view(-1), y)
    loss.backward(gradient=ONE)
    opt.step()

for _ in range(5):
    body()
torch.cuda.synchronize()
with _lib.kernel_timer(capacity=256) as kt:
    body()
torch.cuda.synchronize()
tot = 0.0
for i, (name, grid, us) in enumerate(kt.records()):
    tot += us
    print(f"{i:3d} {us:7.2f} us  grid {grid:6d}  {name}")
print(f"{i + 1} launches, {tot:.1f} us of kernels")
