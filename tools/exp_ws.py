"""Experiment: the warp-specialised fused forward (csrc/triplet_ws.hip, GLAM_FWD_WS) against the barrier-coupled pipelined one and the
general fused kernel: bit equality of the layer output / saved tensors, and per-kernel durations at B = 1024 and B = 16 384."""
import os, sys
os.environ.setdefault("GLAM_TORCH_EXT", "0")     # the Python node: it is the one that takes the ELL routes below the LLC size
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glam_amd import _lib, layer, ops
from glam_amd.data import synth_batch
dev = torch.device("cuda")
sizes = [int(v) for v in sys.argv[1:]] or [3, 64, 1024, 16384]
for B in sizes:
    b = synth_batch(B, seed=B).to(dev)
    torch.manual_seed(0)
    conv = layer.TripletMessage(60, 4).to(dev)
    with torch.no_grad(): conv.bias.normal_(0, 0.1)
    x = torch.randn(b.x.size(0), 60, device=dev)
    res, times = {}, {}
    for name, pf, ws, prod in (("general", "0", "0", "8"), ("pipe", "1", "0", "8"), ("ws4", "1", "1", "4"), ("ws", "1", "1", "8")):
        ops.PIPE_FUSED = pf
        os.environ["GLAM_FWD_WS"] = ws
        os.environ["GLAM_WS_PROD"] = prod
        def run():
            xx = x.clone().requires_grad_(True)
            out = conv(xx, b.edge_index, b.edge_attr)
            g = torch.autograd.grad(out.sum(), [xx] + list(conv.parameters()))
            return out, g
        for _ in range(3): r = run()
        torch.cuda.synchronize()
        with _lib.kernel_timer(capacity=4096) as kt:
            for _ in range(10): run()
        torch.cuda.synchronize()
        acc = {}
        for n, grid, us in kt.records():
            a = acc.setdefault(n, [0.0, 0, grid]); a[0] += us; a[1] += 1
        times[name] = {n: (v[0] / v[1], v[2]) for n, v in acc.items() if "fwd" in n}
        res[name] = r
    eq = {n: (torch.equal(res["general"][0], res[n][0]), all(torch.equal(a, c) for a, c in zip(res["general"][1], res[n][1]))) for n in ("pipe", "ws4", "ws")}
    print(f"B={B} N={b.x.size(0)}: out/grads equal to general: {eq}; max|d out| ws = {(res['general'][0] - res['ws'][0]).abs().max().item():.3g}")
    for n, t in times.items():
        print("   ", n, {k: (round(v[0], 2), v[1]) for k, v in t.items()})
