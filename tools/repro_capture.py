import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.getcwd())
import torch
from glam_amd import model, ops
from glam_amd.data import synth_batch
dev = torch.device("cuda")
torch.manual_seed(0)
net = model.Architecture(mol_block="_TripletMessage", message_steps=3, mol_readout="GlobalPool5", graph_norm="_None",
                         graph_do="_None()", end_do="_None()", pre_act="ReLU", graph_act="ReLU", flat_act="ReLU").to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True, fused=True)
b = synth_batch(32, seed=0).to(dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.mse_loss(net(b).view(-1), b.y.view(-1))
    loss.backward()
    opt.step()
    return loss
mode = sys.argv[1]
ops.GRAD_CARRY = "E1" in mode
if mode.startswith("eager_then_capture"):
    step(); torch.cuda.synchronize(); print("eager ok", flush=True)
    if "Z" in mode:
        opt.zero_grad(set_to_none=True); torch.cuda.synchronize()
ops.GRAD_CARRY = "C1" in mode
if False:
    pass
elif mode == "side_warmup":
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize(); print("side warmup ok", flush=True)
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g):
    l = step()
print("capture ok", flush=True)
g.replay(); torch.cuda.synchronize(); print("replay ok", float(l))
