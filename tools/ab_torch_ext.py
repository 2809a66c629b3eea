"""Eager (no hipGraph) step time with the Python autograd.Function / ctypes route vs the torch-extension route
(GLAM_TORCH_EXT=1: torch.ops.glam.triplet_layer, C++ autograd node): the bench.py layer and the full model at B = 32 / 1024."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glam_amd import layer, model, ops
from glam_amd.data import synth_batch
dev = torch.device("cuda")

def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

for B in (32, 1024):
    b = synth_batch(B, seed=0).to(dev)
    torch.manual_seed(0)
    conv = layer.TripletMessage(60, 4).to(dev)
    x = torch.randn(b.x.size(0), 60, device=dev, requires_grad=True)
    cot = torch.randn(b.x.size(0), 60, device=dev)
    net = model.Architecture(mol_block="_TripletMessage", graph_do="_None()", end_do="_None()", pre_act="ReLU", graph_act="ReLU", flat_act="ReLU").to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
    y = b.y.view(-1)
    def layer_step():
        out = conv(x, b.edge_index, b.edge_attr)
        torch.autograd.grad(out, [x] + list(conv.parameters()), grad_outputs=cot)
    def model_step():
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.mse_loss(net(b).view(-1), y).backward()
        opt.step()
    for ext in (False, True):
        ops.USE_TORCH_EXT = ext
        print(f"B={B:5d} torch_ext={int(ext)}: layer fwd+bwd {timeit(layer_step):7.1f} us/step eager   full model step {timeit(model_step, 100):8.1f} us eager")

# leanest eager step: the extension operator called directly (no Module / propagate / ops dispatch in between)
from glam_amd import torch_ext
G = torch_ext.load()
b = synth_batch(1024, seed=0).to(dev)
N = b.x.size(0)
conv = layer.TripletMessage(60, 4).to(dev)
x = torch.randn(N, 60, device=dev, requires_grad=True)
cot = torch.randn(N, 60, device=dev)
gi = ops.graph_index(b.edge_index, N)
colptr, dst, eid_t = gi.transpose()
ps = list(conv.parameters())
args = (b.edge_attr, *ps, gi.rowptr, gi.src, gi.eid, colptr, dst, eid_t, 3, 0.2)
tl = G.triplet_layer
def lean():
    out = tl(x, *args)
    torch.autograd.grad(out, ps + [x], grad_outputs=cot)
print(f"B= 1024 direct torch.ops.glam.triplet_layer fwd+bwd: {timeit(lean, 500):7.1f} us/step eager")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): lean()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
with torch.cuda.graph(g): lean()
print(f"B= 1024 same step captured in a hipGraph:            {timeit(g.replay, 500):7.1f} us/step")
