"""Developer aid: the optimizer launch on a default-shaped model's parameter list (run under rocprofv3 --kernel-trace for the kernel time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glam_amd import model, optim
dev = torch.device("cuda:0")
net = model.Architecture(mol_block="_TripletMessage").to(dev)
which = sys.argv[1] if len(sys.argv) > 1 else "glam"
opt = optim.Adam(net.parameters(), lr=1e-3) if which == "glam" else torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True, fused=True)
for p in net.parameters():
    p.grad = torch.randn_like(p)
for _ in range(5):
    opt.step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20):
        opt.step()
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    g.replay()
torch.cuda.synchronize()
print(f"{which}: {(time.perf_counter() - t0) / 1000 * 1e6:.2f} us per optimizer step (graph of 20), {sum(p.numel() for p in net.parameters())} parameters in {len(list(net.parameters()))} tensors")
