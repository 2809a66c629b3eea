"""Randomised sweep of glam_dense_gemm (all four layout combinations, gate, bias, activation, all-ones column, unaligned shapes) and
of the linear pair glam_linear_dense_fwd / _bwd against fp64, then the timing of the readout MLP's three products beside torch's
(GEMM library) ones.  usage: fuzz_dense.py [n] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from glam_amd import _lib
dev = torch.device("cuda")
lib, p = _lib.load(), _lib.ptr
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0


def pick(vals):
    return int(rng.choice(vals))


for case in range(n_cases):
    try:
        R, Cn, K = pick([1, 5, 31, 32, 64, 75, 100, 257, 300, 617, 1024, 1100]), pick([1, 2, 12, 60, 64, 75, 300, 301, 617, 1024]), \
            pick([4, 7, 31, 32, 33, 75, 150, 300, 450, 1024, 2039])
        a_kc, b_kc = bool(rng.random() < 0.5), bool(rng.random() < 0.5)
        gate, bias = bool(rng.random() < 0.5), bool(rng.random() < 0.5)
        act = pick([0, 1, 2])
        ones = (not b_kc) and Cn > 1 and bool(rng.random() < 0.5)
        A = torch.randn(R, K)
        G = torch.randn(R, K)
        B = torch.randn(K, Cn)
        bv = torch.randn(Cn)
        gs, sl = float(rng.random()) * 0.5, float(rng.random()) * 0.5
        Ag = A.double() * torch.where(G > 0, 1.0, gs).double() if gate else A.double()
        ref = Ag @ B.double()
        if bias:
            ref = ref + bv.double()
        if act == 1:
            ref = ref.clamp_min(0)
        elif act == 2:
            ref = torch.where(ref > 0, ref, ref * sl)
        Ad = (A if a_kc else A.t().contiguous()).to(dev)
        Gd = (G if a_kc else G.t().contiguous()).to(dev)
        Bd = (B.t().contiguous() if b_kc else B).to(dev)
        bd = bv.to(dev)
        ldc = Cn + pick([0, 0, 1, 4])
        C = torch.full((R, ldc), float("nan"), device=dev)
        rs = torch.full((R,), float("nan"), device=dev)
        rc = lib.glam_dense_gemm(p(Ad), K if a_kc else 1, 1 if a_kc else R, p(Gd) if gate else None, gs, p(Bd), 1 if b_kc else Cn,
                                 K if b_kc else 1, p(bd) if bias else None, act, sl, p(C), ldc, p(rs) if ones else None, R, Cn, K,
                                 _lib.stream())
        assert rc == 0, lib.glam_last_error()
        got = C[:, :Cn].cpu().double()
        scale = max(1.0, ref.abs().max().item())
        err = (got - ref).abs().max().item() / scale
        assert err < 2e-6 * max(1.0, K ** 0.5 / 8), f"R={R} Cn={Cn} K={K} a_kc={a_kc} b_kc={b_kc} gate={gate} bias={bias} act={act}: {err:.2e}"
        assert ldc == Cn or torch.isnan(C[:, Cn:]).all(), "wrote beyond Cn"
        if ones:
            e2 = (rs.cpu().double() - Ag.sum(1)).abs().max().item() / max(1.0, Ag.sum(1).abs().max().item())
            assert e2 < 2e-6 * max(1.0, K ** 0.5 / 8), f"rowsum R={R} K={K}: {e2:.2e}"
    except Exception as e:   # noqa: BLE001
        bad += 1
        print("FAIL case", case, "->", type(e).__name__, str(e)[:300], flush=True)
print(f"{n_cases - bad}/{n_cases} cases passed", flush=True)


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)


for (N, K, M) in [(1024, 300, 1024), (642, 300, 1024), (2039, 300, 1024), (32, 300, 1024), (1024, 450, 1024), (1024, 1024, 617)]:
    x, w, b = torch.randn(N, K, device=dev), torch.randn(M, K, device=dev) * 0.05, torch.randn(M, device=dev)
    y = torch.empty(N, M, device=dev)
    dy = torch.randn(N, M, device=dev)
    dx, dw, db = torch.empty(N, K, device=dev), torch.empty(M, K, device=dev), torch.empty(M, device=dev)
    t_f = timed(lambda: lib.glam_linear_dense_fwd(p(x), p(w), p(b), N, K, M, 1, 0.0, p(y), _lib.stream()))
    ref = torch.relu(torch.addmm(b, x, w.t()))
    ef = ((y - ref).abs().max() / ref.abs().max()).item()
    t_b = timed(lambda: lib.glam_linear_dense_bwd(p(x), p(w), p(dy), p(y), 0.0, N, K, M, p(dx), p(dw), p(db), _lib.stream()))
    g = dy * (y > 0)
    eb = max(((dx - g @ w).abs().max() / (g @ w).abs().max()).item(), ((dw - g.t() @ x).abs().max() / (g.t() @ x).abs().max()).item(),
             ((db - g.sum(0)).abs().max() / g.sum(0).abs().max()).item())
    t_lf = timed(lambda: torch.relu(torch.addmm(b, x, w.t())))
    t_lb = timed(lambda: ((dy * (y > 0)) @ w, (dy * (y > 0)).t() @ x, dy.sum(0)))
    print(f"N={N} K={K} M={M}: forward {t_f:.2f} us (library addmm + relu {t_lf:.2f}), backward pair {t_b:.2f} us (library 2 mm + mask + sum "
          f"{t_lb:.2f}); max rel err vs the library fwd {ef:.1e} bwd {eb:.1e}", flush=True)
sys.exit(1 if bad else 0)
