"""Timing of the readout MLP's linear (glam_linear_dense_fwd / _bwd, hipGraph replay) beside the GEMM library's products.
usage: bench_dense.py [--lib-only-ours]   (GLAM_HIP_LIB selects an experimental build)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glam_amd import _lib
dev = torch.device("cuda")
lib, p = _lib.load(), _lib.ptr
ours_only = "--ours" in sys.argv


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
    t_b = timed(lambda: lib.glam_linear_dense_bwd(p(x), p(w), p(dy), p(y), 0.0, N, K, M, p(dx), p(dw), p(db), _lib.stream()))
    t_w = timed(lambda: lib.glam_linear_dense_bwd(p(x), p(w), p(dy), p(y), 0.0, N, K, M, None, p(dw), p(db), _lib.stream()))
    t_x = timed(lambda: lib.glam_linear_dense_bwd(p(x), p(w), p(dy), p(y), 0.0, N, K, M, p(dx), None, None, _lib.stream()))
    ws = torch.zeros(lib.glam_dense_ws_bytes(), dtype=torch.uint8, device=dev)
    t_fs = timed(lambda: lib.glam_linear_dense_fwd_ws(p(x), p(w), p(b), N, K, M, 1, 0.0, p(y), p(ws), ws.numel(), _lib.stream()))
    t_bs = timed(lambda: lib.glam_linear_dense_bwd_ws(p(x), p(w), p(dy), p(y), 0.0, N, K, M, p(dx), p(dw), p(db), p(ws), ws.numel(), _lib.stream()))
    line = f"N={N} K={K} M={M}: forward {t_f:.2f} us, backward pair {t_b:.2f} us (dw alone {t_w:.2f}, dx alone {t_x:.2f}); with the split-k workspace {t_fs:.2f} / {t_bs:.2f}"
    if not ours_only:
        t_lf = timed(lambda: torch.relu(torch.addmm(b, x, w.t())))
        t_lb = timed(lambda: ((dy * (y > 0)) @ w, (dy * (y > 0)).t() @ x, dy.sum(0)))
        line += f"; library: addmm + relu {t_lf:.2f}, 2 mm + mask + sum {t_lb:.2f}"
    print(line, flush=True)
