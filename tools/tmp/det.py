import sys, os
sys.path.insert(0, ".")
import torch
from glam_amd import _lib
dev = torch.device("cuda")
lib, p = _lib.load(), _lib.ptr
for (N, K, M) in [(64, 300, 1024), (642, 300, 1024), (2039, 300, 1024), (1024, 300, 1024)]:
    torch.manual_seed(N)
    x, w, b = torch.randn(N, K, device=dev), torch.randn(M, K, device=dev) * 0.05, torch.randn(M, device=dev)
    dy = torch.randn(N, M, device=dev)
    first = None
    nbad = 0
    for it in range(60):
        y = torch.full((N, M), float("nan"), device=dev)
        dx, dw, db = torch.full((N, K), float("nan"), device=dev), torch.full((M, K), float("nan"), device=dev), torch.full((M,), float("nan"), device=dev)
        assert lib.glam_linear_dense_fwd(p(x), p(w), p(b), N, K, M, 0, 0.0, p(y), _lib.stream()) == 0
        assert lib.glam_linear_dense_bwd(p(x), p(w), p(dy), None, 0.0, N, K, M, p(dx), p(dw), p(db), _lib.stream()) == 0
        torch.cuda.synchronize()
        cur = [t.clone() for t in (y, dx, dw, db)]
        if first is None:
            first = cur
            ref = [torch.addmm(b.double(), x.double(), w.double().t()), dy.double() @ w.double(), dy.double().t() @ x.double(), dy.double().sum(0)]
            print(N, "err vs fp64:", [f"{((a.double() - r).abs().max() / r.abs().max()).item():.1e}" for a, r in zip(cur, ref)])
        else:
            for name, a, f in zip(("y", "dx", "dw", "db"), cur, first):
                if not torch.equal(a, f):
                    nbad += 1
                    d = (a - f).abs()
                    print(f"  N={N} run {it}: {name} differs: max {d.max().item():.3e} at {int(d.argmax())} nan={int(torch.isnan(a).sum())}")
    print(N, "nondeterministic runs:", nbad)
print("library fp32 for comparison:")
for (N, K, M) in [(64, 300, 1024), (2039, 300, 1024)]:
    torch.manual_seed(N)
    x, w, b = torch.randn(N, K, device=dev), torch.randn(M, K, device=dev) * 0.05, torch.randn(M, device=dev)
    dy = torch.randn(N, M, device=dev)
    cur = [torch.addmm(b, x, w.t()), dy @ w, dy.t() @ x, dy.sum(0)]
    ref = [torch.addmm(b.double(), x.double(), w.double().t()), dy.double() @ w.double(), dy.double().t() @ x.double(), dy.double().sum(0)]
    print(N, "lib err vs fp64 (max):", [f"{((a.double() - r).abs().max() / r.abs().max()).item():.1e}" for a, r in zip(cur, ref)],
          "rms:", [f"{((a.double() - r).pow(2).mean().sqrt() / r.abs().max()).item():.1e}" for a, r in zip(cur, ref)])
    y = torch.empty(N, M, device=dev); dx = torch.empty(N, K, device=dev); dw = torch.empty(M, K, device=dev); db = torch.empty(M, device=dev)
    lib.glam_linear_dense_fwd(p(x), p(w), p(b), N, K, M, 0, 0.0, p(y), _lib.stream())
    lib.glam_linear_dense_bwd(p(x), p(w), p(dy), None, 0.0, N, K, M, p(dx), p(dw), p(db), _lib.stream())
    cur = [y, dx, dw, db]
    print(N, "ours err vs fp64 (max):", [f"{((a.double() - r).abs().max() / r.abs().max()).item():.1e}" for a, r in zip(cur, ref)],
          "rms:", [f"{((a.double() - r).pow(2).mean().sqrt() / r.abs().max()).item():.1e}" for a, r in zip(cur, ref)])
