import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from glam_amd import layer
from glam_amd.data import synth_batch
import oracle.glam_oracle as O
dev = torch.device("cuda")
torch.set_num_threads(16)
for B in (16384, 65536):
    b = synth_batch(B, seed=3)
    N = b.x.size(0)
    torch.manual_seed(1)
    conv = layer.TripletMessage(60, 4)
    x0 = torch.randn(N, 60)
    ps0 = [p.detach().clone().requires_grad_(True) for p in conv.parameters()]
    xo = x0.clone().requires_grad_(True)
    t0 = time.time()
    ref = O.triplet_message(xo, b.edge_index, b.edge_attr, *ps0)
    cot = torch.randn(ref.shape)
    g_ref = torch.autograd.grad((ref * cot).sum(), [xo] + ps0)
    t1 = time.time()
    convd = conv.to(dev)
    x = x0.to(dev).requires_grad_(True)
    out = convd(x, b.edge_index.to(dev), b.edge_attr.to(dev))
    gs = torch.autograd.grad((out * cot.to(dev)).sum(), [x] + list(convd.parameters()))
    torch.cuda.synchronize()
    worst = (out.cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    gw = max((a.cpu() - r).abs().max().item() / max(1.0, r.abs().max().item()) for a, r in zip(gs, g_ref))
    print(f"B={B} N={N} E={b.edge_index.size(1)}: out rel err {worst:.2e}, worst grad rel err {gw:.2e} (oracle {t1 - t0:.1f} s)", flush=True)
    assert worst < 2e-5 and gw < 2e-4
