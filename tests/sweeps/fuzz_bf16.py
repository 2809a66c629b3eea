"""bf16 row-storage mode against the fp32 oracle at bf16 tolerance over random graphs (the exact storage-model check lives in
tests/test_gpu_parity.py::test_triplet_bf16_row_storage).  usage: python tests/sweeps/fuzz_bf16.py [n] [seed]"""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from glam_amd import layer, ops
from glam_amd.data import synth_batch, synth_protein_batch
import oracle.glam_oracle as O
dev = torch.device("cuda")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    C, H, De = int(rng.choice([45, 60, 48, 64, 36])), int(rng.choice([1, 2, 3])), int(rng.choice([4, 8]))
    if H * ((C + 3) // 4 * 4) + 8 > 192:
        H = 2
    prot = rng.random() < 0.3
    b = synth_protein_batch(int(rng.integers(1, 4)), seed=case, n_min=20, n_max=150) if prot else synth_batch(int(rng.integers(1, 40)), seed=case)
    desc = f"case {case}: C={C} H={H} De={De} N={b.x.size(0)}"
    try:
        torch.manual_seed(case)
        N, E = b.x.size(0), b.edge_index.size(1)
        x0 = torch.randn(N, C)
        ea = torch.rand(E, De) if (prot or De != 4) else b.edge_attr
        conv = layer.TripletMessage(C, De, heads=H)
        ps0 = [p.detach().clone().requires_grad_(True) for p in conv.parameters()]
        xo = x0.clone().requires_grad_(True)
        ref = O.triplet_message(xo, b.edge_index, ea, *ps0, heads=H)
        cot = torch.randn(ref.shape)
        g_ref = torch.autograd.grad((ref * cot).sum(), [xo] + ps0)
        convd = copy.deepcopy(conv).to(dev)
        x = x0.to(dev).requires_grad_(True)
        with ops.feature_storage("bf16"):
            out = convd(x, b.edge_index.to(dev), ea.to(dev))
        gs = torch.autograd.grad((out * cot.to(dev)).sum(), [x] + list(convd.parameters()))
        e = (out.cpu() - ref).abs().max().item() / max(1e-6, ref.abs().max().item())
        assert e < 1.5e-2, f"out {e:.2e}"
        for a, r in zip(gs, g_ref):
            ge = (a.cpu() - r).abs().max().item() / max(1e-6, r.abs().max().item())
            assert ge < 3e-2, f"grad {ge:.2e}"
        print("ok  ", desc, f"(out {e:.1e})", flush=True)
    except Exception as ex:   # noqa: BLE001
        bad += 1
        print("FAIL", desc, "->", type(ex).__name__, str(ex)[:160], flush=True)
print(f"{n_cases - bad}/{n_cases} cases passed")
sys.exit(1 if bad else 0)
