"""Developer aid: cycle stamps of k_ts_gemm blocks (library built with -DGLAM_TS_PROF, pointed to by GLAM_HIP_LIB)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from glam_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
N = 20400
for K, M in [(64, 188), (188, 64)]:
    A = torch.randn(N, K, device=dev)
    W = torch.randn(K, M, device=dev)
    img = torch.empty(lib.glam_ts_gemm_image_bytes(K, M) // 4, device=dev)
    _lib.check(lib.glam_ts_gemm_make_image(_lib.ptr(W), M, 0, K, M, _lib.ptr(img), _lib.stream()), "img")
    out = torch.empty(N, M, device=dev)
    for _ in range(5):
        _lib.check(lib.glam_ts_gemm(_lib.ptr(A), K, K, None, 0, 0, _lib.ptr(img), None, _lib.ptr(out), M, M, None, 0, 0, N, _lib.stream()), "gemm")
    torch.cuda.synchronize()
    nb = 256
    buf = (ctypes.c_longlong * (nb * 8))()
    assert raw.glam_debug_ts_prof(buf, nb * 8) == 0
    st = np.array(buf[:], dtype=np.int64).reshape(nb, 8)[:, :6]
    st = st[st[:, 0] > 0]
    d = np.diff(st, axis=1)
    print(f"K={K} M={M} blocks={len(st)}")
    for i, n in enumerate(["A loads issued + image staged + barrier", "drain (A fragment arrival)", "MFMAs of the first item", "rest (stores issued, later items)", "drain stores"]):
        print(f"   {n:42s} mean {d[:, i].mean():8.0f}  max {d[:, i].max():8.0f}")
    print("   block lifetime mean", (st[:, 5] - st[:, 0]).mean(), " first start -> last end:", st[:, 5].max() - st[:, 0].min(), " start spread:", st[:, 0].max() - st[:, 0].min())
