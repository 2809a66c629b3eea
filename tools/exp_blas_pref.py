import torch, time, sys
dev = torch.device("cuda")
x = torch.randn(1024, 300, device=dev); w = torch.randn(1024, 300, device=dev); b = torch.randn(1024, device=dev)
x32 = torch.randn(32, 300, device=dev)
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for pref in ("default", "hipblaslt"):
    try:
        torch.backends.cuda.preferred_blas_library(pref)
    except Exception as e:
        print(pref, "unavailable", e); continue
    print(pref, "B=1024 linear 300->1024: %.1f us" % t(lambda: torch.nn.functional.linear(x, w, b)),
          " B=32: %.1f us" % t(lambda: torch.nn.functional.linear(x32, w, b)),
          " dW (1024x300 = dy^T x): %.1f us" % t(lambda: torch.matmul(torch.randn(1024,1024,device=dev).t() if False else x.t(), x)))
