import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from glam_amd import layer
from glam_amd.data import synth_batch
dev = torch.device("cuda:0")
b = synth_batch(1024, seed=0).to(dev)
torch.manual_seed(0)
conv = layer.TripletMessage(60, 4).to(dev)
x = torch.randn(b.x.size(0), 60, device=dev, requires_grad=True)
cot = torch.randn(b.x.size(0), 60, device=dev)
params = list(conv.parameters())
def body():
    out = conv(x, b.edge_index, b.edge_attr)
    return torch.autograd.grad(out, params + [x], grad_outputs=cot)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): body()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
for S in (1, 2, 4, 8, 16):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = [body() for _ in range(S)]
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    reps = 4000 // S
    t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"steps per graph {S:2d}: {dt / (reps * S) * 1e6:.2f} us/step", flush=True)
