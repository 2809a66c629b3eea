#!/usr/bin/env python3
"""Per-launch time of the warp-specialised GRU step kernels against the row count: python3 tools/bench_gru.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glam_amd import _lib
lib, p = _lib.load(), _lib.ptr
dev = torch.device("cuda")
st = lambda: torch.cuda.current_stream().cuda_stream
C = int(os.environ.get("C", "60")); M = 3 * C
Ns = (1024, 5120, 20400, 81920, 326400)
def timeit(go):
    for _ in range(5): go()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): go()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 200 * 1e6
print(f"C={C}   " + " ".join(f"{n:>9}" for n in Ns) + "   (us per launch)")
rows = {"gru_ws_fwd": [], "gru_ws_fwd_pre": [], "gru_fused_fwd (fp32)": [], "gru_bwd_ws": [], "gru_bwd_ws_pre": [], "tail_bwd + pair (x3)": []}
for N in Ns:
    r = lambda *s: torch.randn(*s, device=dev)
    x, h, idn, w_ih, w_hh, b_ih, b_hh = r(N, C), r(N, C), r(N, C), r(M, C) * 0.3, r(M, C) * 0.3, r(M), r(M)
    ia, ib = (torch.empty(lib.glam_ts_gemm_image_bytes(C, M) // 4, device=dev) for _ in range(2))
    ta, tb = (torch.empty(lib.glam_ts_gemm_image_bytes(M, C) // 4, device=dev) for _ in range(2))
    for w, i, t in ((w_ih, ia, ta), (w_hh, ib, tb)):
        assert lib.glam_ts_gemm_make_image(p(w), C, 1, C, M, p(i), st()) == 0
        assert lib.glam_ts_gemm_make_image(p(w), C, 0, M, C, p(t), st()) == 0
    fz = torch.empty(2, lib.glam_gru_fused_image_bytes() // 4, device=dev)
    assert lib.glam_gru_fused_make_images(p(w_ih), p(w_hh), C, p(fz[0]), p(fz[1]), st()) == 0
    gi, gh, hn, out = torch.empty(N, M, device=dev), torch.empty(N, M, device=dev), torch.empty(N, C, device=dev), torch.empty(N, C, device=dev)
    rows["gru_ws_fwd"].append(timeit(lambda: lib.glam_gru_ws_fwd(p(x), p(h), p(idn), p(ia), p(ib), p(b_ih), p(b_hh), N, C, 1, 1, 0.0, p(gi), p(gh), p(hn), p(out), st())))
    pre = torch.empty(2, lib.glam_gru_ws_pre_bytes(), dtype=torch.uint8, device=dev)
    assert lib.glam_gru_ws_make_pre(p(w_ih), p(w_hh), C, p(pre[0]), p(pre[1]), st()) == 0
    rows["gru_ws_fwd_pre"].append(timeit(lambda: lib.glam_gru_ws_fwd_pre(p(x), p(h), p(idn), p(pre[0]), p(b_ih), p(b_hh), N, C, 1, 1, 0.0, p(gi), p(gh), p(hn), p(out), None, st())))
    rows["gru_fused_fwd (fp32)"].append(timeit(lambda: lib.glam_gru_fused_fwd(p(x), p(h), p(idn), p(fz[0]), p(fz[1]), p(b_ih), p(b_hh), N, C, 1, 1, 0.0, p(gi), p(gh), p(hn), p(out), st())))
    d_out, d_hs = r(N, C), r(N, C)
    dgi, dgh, did, dx, dh, dh2 = (torch.empty(N, M, device=dev), torch.empty(N, M, device=dev), torch.empty(N, C, device=dev), torch.empty(N, C, device=dev),
                                  torch.empty(N, C, device=dev), torch.empty(N, C, device=dev))
    rows["gru_bwd_ws"].append(timeit(lambda: lib.glam_gru_bwd_ws(p(gi), p(gh), p(h), p(out), p(d_out), p(d_hs), p(x), p(ta), p(tb), N, C, 1, 1, 0.0, 0, p(dgi), p(dgh), p(did), p(dx), p(dh), st())))
    rows["gru_bwd_ws_pre"].append(timeit(lambda: lib.glam_gru_bwd_ws_pre(p(gi), p(gh), p(h), p(out), p(d_out), p(d_hs), p(x), p(pre[1]), N, C, 1, 1, 0.0, 0, p(dgi), p(dgh), p(did), p(dx), p(dh), st())))
    def two():
        lib.glam_gru_tail_bwd(p(gi), p(gh), p(h), p(out), p(d_out), p(d_hs), N, C, 1, 0.0, p(dgi), p(dgh), p(dh), p(did), st())
        lib.glam_ts_gemm_pair(p(dgi), M, M, 0, p(ta), None, p(dx), C, C, p(x), C, None, 0, p(dgh), M, M, 0, p(tb), None, p(dh2), C, C, None, 0, p(dh), C, N, st())
    rows["tail_bwd + pair (x3)"].append(timeit(two))
for k, v in rows.items():
    print(f"{k:22s}" + " ".join(f"{t:9.2f}" for t in v))
