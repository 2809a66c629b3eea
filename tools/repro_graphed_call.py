import os, sys, copy
sys.path.insert(0, "/root/repo")
import torch
from glam_amd import model, graphs
from glam_amd.data import synth_batch
V = os.environ.get("V", "0")
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = model.Architecture(mol_block="_TripletMessage", message_steps=2, mol_readout="GlobalPool5", e_dim=64, graph_norm="_None", graph_do="_None()",
                         end_do="_None()", pre_act="ReLU", graph_act="ReLU", flat_act="ReLU").to(dev)
net.graphed_call = False
b = synth_batch(8, seed=1).to(dev)
out = net(b); keep = out.sum(); keep.backward(); net.zero_grad(set_to_none=True); del out
if V not in ("4", "5"):
    del keep                 # 4 / 5: the caller still holds last step's loss (the reference's loop does) — its AccumulateGrad nodes stay alive
params = tuple(net.parameters())
names = [n for n, _ in net.named_parameters()]
if V == "5":
    proxies = tuple(p.detach().requires_grad_() for p in params)
    run = lambda: torch.func.functional_call(net, dict(zip(names, proxies)), (b,))
    params = proxies
else:
    run = lambda: net(b)
torch.cuda.synchronize()
fwd = torch.cuda.CUDAGraph()
with torch.cuda.graph(fwd):
    out = run()
    if V == "1":
        gout = torch.zeros_like(out)
print("fwd captured", flush=True)
if V != "1":
    gout = torch.zeros_like(out)
bwd = torch.cuda.CUDAGraph()
kw = dict(pool=fwd.pool()) if V != "3" else {}
with torch.cuda.graph(bwd, **kw):
    grads = torch.autograd.grad(out, params, gout, allow_unused=True, retain_graph=(V == "2"))
print("bwd captured", flush=True)
fwd.replay(); gout.fill_(1.0); bwd.replay(); torch.cuda.synchronize()
print("ok", V, float(grads[0].abs().sum()))
