#!/usr/bin/env python3
"""Per-launch time of the tall-skinny products against the row count (fixed cost vs streaming rate): python3 tools/bench_ts_gemm.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glam_amd import _lib
lib, p = _lib.load(), _lib.ptr
dev = torch.device("cuda")
st = lambda: torch.cuda.current_stream().cuda_stream
shapes = [(92, 284), (276, 92), (180, 60), (60, 188)]
print(f"{'K':>4} {'M':>4} " + " ".join(f"{n:>9}" for n in (1024, 5120, 20400, 81920, 326400)) + "   (us per launch; GB/s at the largest N)")
for K, M in shapes:
    row = []
    for N in (1024, 5120, 20400, 81920, 326400):
        A = torch.randn(N, K, device=dev); W = torch.randn(K, M, device=dev); out = torch.empty(N, M, device=dev)
        img = torch.empty(lib.glam_ts_gemm_image_bytes(K, M) // 4, device=dev)
        assert lib.glam_ts_gemm_make_image(p(W), M, 0, K, M, p(img), st()) == 0
        def go():
            assert lib.glam_ts_gemm(p(A), K, K, None, 0, 0, p(img), None, p(out), M, M, None, 0, 0, N, st()) == 0
        for _ in range(5): go()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): go()
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / 200 * 1e6)
    gbs = 326400 * (K + M) * 4 / row[-1] / 1e3
    print(f"{K:>4} {M:>4} " + " ".join(f"{t:9.2f}" for t in row) + f"   {gbs:7.0f}")
