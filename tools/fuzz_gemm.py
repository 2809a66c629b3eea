"""Randomised sweep of the dense C-ABI entry points against fp64: glam_ts_gemm (all variants, split operands / outputs,
bias, transposed weights) and glam_wgrad_gemm (ones columns, chunked J, both stride orders).  usage: fuzz_gemm.py [n] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from glam_amd import _lib
dev = torch.device("cuda")
lib, p = _lib.load(), _lib.ptr
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
def m4(lo, hi): return int(rng.integers(lo // 4, hi // 4 + 1)) * 4
for case in range(n_cases):
    try:
        N = int(rng.choice([1, 3, 15, 16, 17, 100, 1000, 4097, 5000, 20400, 70001]))
        if rng.random() < 0.5:
            # ---- ts_gemm ----
            variant = int(rng.integers(0, 4))        # 3: the long-reduction class beyond the fp32 table (tall_x3.hip)
            K, M = [(m4(4, 192), m4(4, 64)), (m4(4, 64), m4(4, 192)), (m4(4, 96), m4(4, 320)), (m4(196, 288), m4(4, 96))][variant]
            K2 = m4(0, K - 4) if K > 4 and rng.random() < 0.4 else 0
            K1 = K - K2
            M2 = m4(4, M - 4) if M > 4 and rng.random() < 0.4 else 0
            M1 = M - M2
            trans, bias = int(rng.random() < 0.5), bool(rng.random() < 0.5)
            A1, A2 = torch.randn(N, K1), torch.randn(N, max(K2, 1))[:, :K2].contiguous()
            W = torch.randn(K, M)
            b = torch.randn(M1) if bias else None
            ref = torch.cat([A1, A2], 1).double() @ W.double()
            if bias:
                ref[:, :M1] += b.double()
            Wd = (W.t().contiguous() if trans else W).to(dev)
            img = torch.empty(lib.glam_ts_gemm_image_bytes(K, M) // 4, device=dev)
            assert lib.glam_ts_gemm_make_image(p(Wd), K if trans else M, trans, K, M, p(img), _lib.stream()) == 0, lib.glam_last_error()
            o1 = torch.full((N, M1), float("nan"), device=dev)
            o2 = torch.full((N, max(M2, 1)), float("nan"), device=dev)
            A1d, A2d, bd = A1.to(dev), A2.to(dev), (b.to(dev) if bias else None)
            rc = lib.glam_ts_gemm(p(A1d), K1, K1, p(A2d) if K2 else None, K2, K2, p(img), p(bd), p(o1), M1, M1, p(o2) if M2 else None, M2,
                                  max(M2, 4), N, _lib.stream())
            assert rc == 0, lib.glam_last_error()
            got = torch.cat([o1, o2[:, :M2]], 1).cpu().double()
            err = (got - ref).abs().max().item() / max(1.0, ref.abs().max().item())
            assert err < 3e-6, f"ts_gemm K={K1}+{K2} M={M1}+{M2} N={N} trans={trans} bias={bias}: {err:.2e}"
        else:
            # ---- wgrad ----
            I1 = m4(4, 300)
            I2 = m4(0, min(16, 316 - I1)) if rng.random() < 0.3 else 0
            ones = int(rng.random() < 0.5 and I1 + I2 + 1 <= 320)
            J = m4(4, 128)
            qones = int(rng.random() < 0.4 and (J + 1 <= 64 or (J > 64 and J + 1 <= 128)))
            P1, P2, Q = torch.randn(N, I1), torch.randn(N, max(I2, 1))[:, :I2].contiguous(), torch.randn(N, J)
            P = torch.cat([P1, P2] + ([torch.ones(N, 1)] if ones else []), 1)
            Qf = torch.cat([Q] + ([torch.ones(N, 1)] if qones else []), 1)
            ref = P.double().t() @ Qf.double()
            I, Jt = P.size(1), Qf.size(1)
            ws = torch.empty(lib.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=dev)
            transposed = bool(rng.random() < 0.5)
            out = torch.full((Jt, I) if transposed else (I, Jt), float("nan"), device=dev)
            si, sj = (1, I) if transposed else (Jt, 1)
            P1d, P2d, Qd = P1.to(dev), P2.to(dev), Q.to(dev)
            rc = lib.glam_wgrad_gemm(p(P1d), I1, I1, p(P2d) if I2 else None, I2, I2, ones, p(Qd), J, J, qones, N, p(out), si, sj, p(ws),
                                     ws.numel(), _lib.stream())
            assert rc == 0, lib.glam_last_error()
            got = (out.t() if transposed else out).cpu().double()
            err = (got - ref).abs().max().item() / max(1.0, ref.abs().max().item())
            assert err < 5e-6 * max(1.0, N ** 0.5 / 10), f"wgrad I={I1}+{I2}+{ones} J={J}+{qones} N={N} T={transposed}: {err:.2e}"
    except Exception as e:   # noqa: BLE001
        bad += 1
        print("FAIL case", case, "->", type(e).__name__, str(e)[:200], flush=True)
print(f"{n_cases - bad}/{n_cases} cases passed")
sys.exit(1 if bad else 0)
