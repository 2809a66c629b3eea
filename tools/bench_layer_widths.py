"""One TripletMessage layer fwd+bwd at the widths of the reference's search space (hid_dim_alpha in {1,2,3,4,6}),
B=1024, hipGraph replay."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glam_amd import layer
from glam_amd.data import synth_batch

dev = torch.device("cuda:0")
b = synth_batch(1024, seed=0).to(dev)
for alpha in ([int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6]):
    C = 15 * alpha
    torch.manual_seed(0)
    conv = layer.TripletMessage(C, 4).to(dev)
    x = torch.randn(b.x.size(0), C, device=dev, requires_grad=True)
    cot = torch.randn(b.x.size(0), C, device=dev)
    params = list(conv.parameters())
    def body():
        out = conv(x, b.edge_index, b.edge_attr)
        return torch.autograd.grad(out, params + [x], grad_outputs=cot)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): body()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = body()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 5
    print(f"hid_dim_alpha={alpha} C={C}: {us:.1f} us/step  {1024 / us:.2f} M mol/s", flush=True)
