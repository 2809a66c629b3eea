import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from glam_amd import layer
from glam_amd.data import synth_batch
import oracle.glam_oracle as O
dev = torch.device("cuda")
b = synth_batch(12, seed=2)
N = b.x.size(0)
for C, H in [(100, 3), (104, 3), (128, 2), (128, 1), (160, 1), (200, 1), (256, 1), (130, 2), (64, 4), (80, 4)]:
    try:
        torch.manual_seed(C)
        conv = layer.TripletMessage(C, 4, heads=H)
        x0 = torch.randn(N, C)
        ps0 = [p.detach().clone().requires_grad_(True) for p in conv.parameters()]
        xo = x0.clone().requires_grad_(True)
        ref = O.triplet_message(xo, b.edge_index, b.edge_attr, *ps0, heads=H)
        cot = torch.randn(ref.shape)
        g_ref = torch.autograd.grad((ref * cot).sum(), [xo] + ps0)
        convd = conv.to(dev)
        x = x0.to(dev).requires_grad_(True)
        out = convd(x, b.edge_index.to(dev), b.edge_attr.to(dev))
        gs = torch.autograd.grad((out * cot.to(dev)).sum(), [x] + list(convd.parameters()))
        e = (out.cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        ge = max((a.cpu() - r).abs().max().item() / max(1.0, r.abs().max().item()) for a, r in zip(gs, g_ref))
        print(f"C={C} H={H}: out {e:.1e} grads {ge:.1e}", "OK" if e < 3e-5 and ge < 3e-4 else "MISMATCH", flush=True)
    except Exception as ex:
        print(f"C={C} H={H}: {type(ex).__name__}: {str(ex)[:140]}", flush=True)
