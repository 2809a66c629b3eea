import sys, os
sys.path.insert(0, os.getcwd())
import torch
from glam_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda")
N = 20700
for K, M in [(92, 284), (92, 276), (60, 188), (180, 60)]:   # (276, 92) / (284, 92): no kernel variant (library GEMM wins)
    A = torch.randn(N, K, device=dev); W = torch.randn(K, M, device=dev)
    img = ops._ts_image(W, K, M, False)
    out = torch.empty(N, M, device=dev)
    def f(): _lib.check(lib.glam_ts_gemm(_lib.ptr(A), K, K, None, 0, 0, _lib.ptr(img), None, _lib.ptr(out), M, M, None, 0, 0, N, _lib.stream()), "g")
    def g(): torch.matmul(A, W, out=out)
    for fn, nm in ((f, "ts_gemm"), (g, "library")):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"K={K} M={M} {nm}: {e0.elapsed_time(e1)*10:.1f} us", flush=True)
