"""Forward layer time (stage + layer_fwd) on the tile path vs the general path at several batch sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glam_amd import layer, ops
from glam_amd.data import synth_batch

dev = torch.device("cuda:0")
for B in [256, 1024, 2048, 4096, 16384]:
    b = synth_batch(B, seed=0).to(dev)
    conv = layer.TripletMessage(60, 4).to(dev)
    x = torch.randn(b.x.size(0), 60, device=dev)
    gi = ops.graph_index(b.edge_index, x.size(0))
    plan = gi.tile_plan()
    res = {}
    for name, p in [("tile", plan), ("general", None)]:
        ops.TILES_ENABLED = p is not None
        with torch.no_grad():
            for _ in range(5):
                conv(x, b.edge_index, b.edge_attr)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                conv(x, b.edge_index, b.edge_attr)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) * 10
    tp = plan[0].cpu()
    sz = (tp[1:] - tp[:-1])
    print(f"B={B} N={x.size(0)} T={plan[1]} tile nodes max={int(sz.max())} mean={float(sz.float().mean()):.1f}  fwd us: {res}", flush=True)
