import sys
sys.path.insert(0, ".")
import torch
from glam_amd import _lib
dev = torch.device("cuda")
lib, p = _lib.load(), _lib.ptr
torch.manual_seed(0)
N, K, M = 2039, 300, 1024
w = (torch.rand(M, K, device=dev) * 2 - 1) * K ** -0.5
w2 = (torch.rand(2, M, device=dev) * 2 - 1) * M ** -0.5
do = torch.randn(N, 2, device=dev)
for name, dy in (("random dy", torch.randn(N, M, device=dev)), ("rank-2 dy", do @ w2), ("positive dy", torch.rand(N, M, device=dev))):
    x = torch.randn(N, K, device=dev)
    dx, dw, db = torch.empty(N, K, device=dev), torch.empty(M, K, device=dev), torch.empty(M, device=dev)
    lib.glam_linear_dense_bwd(p(x), p(w), p(dy), None, 0.0, N, K, M, p(dx), p(dw), p(db), _lib.stream())
    r = dy.double() @ w.double()
    for tag, got in (("ours", dx), ("lib ", dy @ w)):
        e = got.double() - r
        print(f"{name} dx {tag}: max {e.abs().max().item():.2e} rms {e.pow(2).mean().sqrt().item():.2e} mean {e.mean().item():+.2e} "
              f"mean/rms*sqrt(n) {(e.mean() / e.pow(2).mean().sqrt() * e.numel() ** 0.5).item():+.1f}; colsum err max {(e.sum(0)).abs().max().item():.2e}; "
              f"sign-correlated with value: {((e * r.sign()).mean() / e.abs().mean()).item():+.3f}")
