"""Developer aid: per-dispatch duration of the weight-gradient launch (k_wgrad_x3 or k_wgrad) of the headline step's two products, alone,
at N rows (glam_prof_* timestamps; GLAM_HIP_LIB selects an experimental build).  usage: wx_time.py [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glam_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
p, st = _lib.ptr, _lib.stream
for N in [int(a) for a in sys.argv[1:]] or [20400, 326400]:
    Pa, Qa = torch.randn(N, 180, device=dev), torch.randn(N, 60, device=dev)
    Pb, Qb = torch.randn(N, 188, device=dev), torch.randn(N, 60, device=dev)
    oa, ob = torch.empty(181, 60, device=dev), torch.empty(188, 60, device=dev)
    ws = torch.empty(lib.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=dev)
    def run():
        _lib.check(lib.glam_wgrad_gemm_pair(p(Pa), 180, 180, 1, p(Qa), 60, 60, 0, 0, p(oa), 60, 1, p(Pb), 188, 188, 0, p(Qb), 60, 60, 0, 0, p(ob), 60, 1,
                                            N, p(ws), ws.numel(), st()), "pair")
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    with _lib.kernel_timer(capacity=256) as kt:
        for _ in range(30):
            run()
    torch.cuda.synchronize()
    acc = {}
    for name, grid, us in kt.records():
        acc.setdefault((name, grid), []).append(us)
    ref = (Pa.double().t() @ Qa.double())
    err = float((oa[:180].double() - ref).abs().max() / ref.abs().max())
    print(f"N={N}: " + "  ".join(f"{k[0]}[{k[1]}] {sum(v) / len(v):.2f} us (min {min(v):.2f})" for k, v in acc.items()) + f"  rel.err {err:.1e}")
