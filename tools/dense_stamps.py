"""Clock stamps of one forward launch of glam_linear_dense_fwd (a -DGLAM_DENSE_STAMP build, GLAM_HIP_LIB=...)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glam_amd import _lib
dev = torch.device("cuda")
lib, p = _lib.load(), _lib.ptr
raw = ctypes.CDLL(_lib.LIB_PATH)
N, K, M = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (1024, 300, 1024)
x, w, b = torch.randn(N, K, device=dev), torch.randn(M, K, device=dev) * 0.05, torch.randn(M, device=dev)
y = torch.empty(N, M, device=dev)
for _ in range(5):
    lib.glam_linear_dense_fwd(p(x), p(w), p(b), N, K, M, 1, 0.0, p(y), _lib.stream())
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (8 * 2 * 64))()
assert raw.glam_debug_dense_prof(buf, 8 * 2 * 64) == 0
for blk in (0, 1):
    for grp in (0, 1):
        v = [buf[(blk * 2 + grp) * 64 + k] for k in range(64)]
        t0 = buf[(blk * 2) * 64]
        print(f"block {blk} group {grp}:", " ".join(f"{k}:{v[k] - t0}" for k in range(64) if v[k]))
