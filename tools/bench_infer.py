"""The inference forward (torch.no_grad(): no aggr / stats store) beside the training forward of one TripletMessage(60, 4) layer:
per-dispatch durations (glam_prof_*) and hipGraph-replayed forward-only steps.  usage: bench_infer.py [B ...]"""
import os, sys, time
os.environ.setdefault("GLAM_TORCH_EXT", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glam_amd import _lib, layer, ops
from glam_amd.data import synth_batch

dev = torch.device("cuda:0")
for B in [int(a) for a in sys.argv[1:]] or [1024, 16384]:
    b = synth_batch(B, seed=7 if B > 1024 else 0).to(dev)
    torch.manual_seed(0)
    conv = layer.TripletMessage(60, 4).to(dev)
    N = b.x.size(0)
    x = torch.randn(N, 60, device=dev)
    for infer in (False, True):
        ops.INFER_FWD = infer
        with torch.no_grad(), ops.cached_staging():
            for _ in range(20):
                conv(x, b.edge_index, b.edge_attr)
            torch.cuda.synchronize()
            reps = 30
            with _lib.kernel_timer(capacity=8 * reps) as kt:
                for _ in range(reps):
                    conv(x, b.edge_index, b.edge_attr)
            torch.cuda.synchronize()
            acc = {}
            for name, grid, us in kt.records():
                a = acc.setdefault(name, [0.0, 0])
                a[0] += us; a[1] += 1
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(16):
                    conv(x, b.edge_index, b.edge_attr)
            for _ in range(20):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 200 if B <= 2048 else 30
            for _ in range(n):
                g.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / (16 * n) * 1e6
        print(f"B={B} N={N} {'inference' if infer else 'training '} forward: {dt:7.2f} us/step replayed;  " +
              "  ".join(f"{k} {v[0] / v[1]:.2f} us" for k, v in acc.items()), flush=True)
