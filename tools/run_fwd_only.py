"""Developer aid: nothing but the fused forward of one TripletMessage layer, a few times (the command a rocprofv3 --pmc pass wraps).
usage: run_fwd_only.py B [reps] ; environment: GLAM_WS=0 / GLAM_WS_ROUTE=0 select the general kernels."""
import os, sys
os.environ.setdefault("GLAM_TORCH_EXT", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glam_amd import layer, ops
from glam_amd.data import synth_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
b = synth_batch(B, seed=7).to(dev)
torch.manual_seed(0)
conv = layer.TripletMessage(60, 4).to(dev)
x = torch.randn(b.x.size(0), 60, device=dev)
with torch.no_grad():
    for _ in range(reps):
        conv(x, b.edge_index, b.edge_attr)
torch.cuda.synchronize()
