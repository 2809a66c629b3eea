#!/usr/bin/env python3
"""Developer micro-benchmark of the dense kernels (k_ts_gemm / k_wgrad) over N; prints us per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glam_amd import _lib
lib, p, st = _lib.load(), _lib.ptr, _lib.stream
dev = torch.device("cuda")

def timed(fn, reps=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

for N in [16, 4096, 8192, 16384, 20400, 40960, 81920, 328514]:
    row = [f"N={N:7d}"]
    for (K, M) in [(60, 188), (180, 60), (188, 60)]:
        A = torch.randn(N, K, device=dev); W = torch.randn(K, M, device=dev)
        img = torch.empty(lib.glam_ts_gemm_image_bytes(K, M) // 4, device=dev)
        lib.glam_ts_gemm_make_image(p(W), M, 0, K, M, p(img), st())
        M1 = M - 8 if M == 188 else M; M2 = M - M1
        o1 = torch.empty(N, M1, device=dev); o2 = torch.empty(N, max(M2, 4), device=dev)
        fn = lambda: lib.glam_ts_gemm(p(A), K, K, None, 0, 0, p(img), None, p(o1), M1, M1, p(o2) if M2 else None, M2, 8, N, st())
        t = timed(fn)
        row.append(f"ts {K}x{M}: {t:7.2f}us {2*N*K*M/t/1e6:6.1f}TF")
    ws = torch.empty(lib.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=dev)
    for (I, J) in [(180, 60), (188, 60)]:
        P = torch.randn(N, I, device=dev); Q = torch.randn(N, J, device=dev); out = torch.empty(I + 1, J, device=dev)
        fn = lambda: lib.glam_wgrad_gemm(p(P), I, I, None, 0, 0, 1, p(Q), J, J, 0, N, p(out), J, 1, p(ws), ws.numel(), st())
        t = timed(fn)
        row.append(f"wg {I}x{J}: {t:7.2f}us {2*N*I*J/t/1e6:6.1f}TF")
    print("  ".join(row))
