"""Randomised parity sweep (HIP path vs the CPU oracle) over graph-size mixes, widths, heads and readouts.
usage: python tests/sweeps/fuzz_parity.py [n_cases] [seed]"""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from glam_amd import layer, ops
from glam_amd.data import Batch, Data
import oracle.glam_oracle as O

dev = torch.device("cuda")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)

def rand_graphs(B, lo, hi, De, onehot):
    parts = []
    for _ in range(B):
        n = int(rng.integers(lo, hi + 1))
        m = int(rng.integers(0, 3 * n + 1))
        ei = torch.from_numpy(rng.integers(0, n, size=(2, m))).long() if n > 0 else torch.zeros(2, 0, dtype=torch.long)
        if m and rng.random() < 0.5:
            ei = torch.cat([ei, ei.flip(0)], 1)                      # symmetric, duplicates and self loops allowed
        E = ei.size(1)
        ea = torch.eye(De)[torch.from_numpy(rng.integers(0, De, size=E))] if onehot else torch.rand(E, De)
        parts.append(Data(torch.zeros(n, 1), ei, ea.float()))
    return Batch.from_data_list(parts)

def grads(out, cot, ins):
    return torch.autograd.grad((out * cot).sum(), ins, allow_unused=True)

def close(a, b, tol, what):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs().max().item() if a.numel() else 0.0
    lim = tol * max(1.0, b.abs().max().item() if b.numel() else 1.0)
    assert err <= lim, f"{what}: {err:.3e} > {lim:.3e}"

bad = 0
for case in range(n_cases):
    C = int(rng.choice([15, 30, 45, 60, 90, 32, 64]))
    H = int(rng.choice([1, 2, 3, 3, 3, 4])) if C <= 64 else 3
    De = int(rng.choice([4, 8, 3]))
    B = int(rng.integers(1, 9))
    hi = int(rng.choice([4, 30, 70, 200]))
    b = rand_graphs(B, 0 if rng.random() < 0.3 else 1, hi, De, onehot=bool(rng.random() < 0.5))
    N = b.x.size(0)
    desc = f"case {case}: C={C} H={H} De={De} B={B} N={N} E={b.edge_index.size(1)}"
    try:
        torch.manual_seed(case)
        x0 = torch.randn(N, C)
        conv = layer.TripletMessage(C, De, heads=H)
        with torch.no_grad():
            conv.bias.normal_(0, 0.1)
        ps0 = [p.detach().clone().requires_grad_(True) for p in conv.parameters()]
        xo = x0.clone().requires_grad_(True)
        ref = O.triplet_message(xo, b.edge_index, b.edge_attr, *ps0, heads=H)
        cot = torch.randn(ref.shape)
        g_ref = grads(ref, cot, [xo] + ps0)
        convd = copy.deepcopy(conv).to(dev)
        x = x0.to(dev).requires_grad_(True)
        out = convd(x, b.edge_index.to(dev), b.edge_attr.to(dev))
        close(out, ref, 2e-5, "triplet out")
        for n_, a, r in zip(["x"] + [n for n, _ in convd.named_parameters()], grads(out, cot.to(dev), [x] + list(convd.parameters())), g_ref):
            close(a, r, 1e-4, "triplet grad " + n_)
        if N > 0:
            bd = b.batch.to(dev)
            for name, rf, df in (("pool5", lambda t: O.global_pool5(t, b.batch, B), lambda t: layer.GlobalPool5()(t, bd, B)),
                                 ("pair_norm", lambda t: O.pair_norm(t, b.batch, B), lambda t: ops.pair_norm(t, ops.segment_ptr(bd, B)))):
                xo = x0.clone().requires_grad_(True)
                r = rf(xo); ct = torch.randn(r.shape); (gr,) = grads(r, ct, [xo])
                xd = x0.to(dev).requires_grad_(True)
                o = df(xd)
                close(o, r, 2e-5, name); close(grads(o, ct.to(dev), [xd])[0], gr, 1e-4, name + " d_x")
            ro = layer.Set2Set(C, 3); lref = copy.deepcopy(ro.lstm); ro = ro.to(dev)
            xo = x0.clone().requires_grad_(True)
            r = O.set2set(xo, b.batch, B, lref, steps=3); ct = torch.randn(r.shape); (gr,) = grads(r, ct, [xo])
            xd = x0.to(dev).requires_grad_(True)
            o = ro(xd, bd, B)
            close(o, r, 3e-5, "set2set"); close(grads(o, ct.to(dev), [xd])[0], gr, 1e-4, "set2set d_x")
        print("ok  ", desc, flush=True)
    except Exception as e:   # noqa: BLE001 - report and keep sweeping
        bad += 1
        print("FAIL", desc, "->", type(e).__name__, str(e)[:160], flush=True)
print(f"{n_cases - bad}/{n_cases} cases passed")
sys.exit(1 if bad else 0)
