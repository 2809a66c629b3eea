"""Developer aid: per-phase shader-clock breakdown of k_tile_fwd (needs a library built with -DGLAM_TILE_PROF,
pointed to by GLAM_HIP_LIB)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from glam_amd import _lib, layer, ops
ops.TILES_ENABLED = True
from glam_amd.data import synth_batch

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b = synth_batch(B, seed=0).to(dev)
conv = layer.TripletMessage(60, 4).to(dev)
x = torch.randn(b.x.size(0), 60, device=dev)
with torch.no_grad():
    for _ in range(5):
        conv(x, b.edge_index, b.edge_attr)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
T = (x.size(0) + 79) // 80
buf = (ctypes.c_longlong * (T * 8))()
assert raw.glam_debug_tile_prof(buf, T * 8) == 0
st = np.array(buf[:], dtype=np.int64).reshape(T, 8)[:, :6]
d = np.diff(st, axis=1).astype(np.float64)
names = ["prologue", "phase A (node GEMM)", "img issue", "phase B (aggregate)", "wait+barrier", "phase C (update GEMM)"]
names = ["prologue", "phase A", "phase B (+img issue)", "wait img / barrier", "phase C"]
print(f"tiles={T}  total cycles/tile: mean {d.sum(1).mean():.0f}  max {d.sum(1).max():.0f}")
for i, n in enumerate(names):
    print(f"  {n:24s} mean {d[:, i].mean():8.0f}  min {d[:, i].min():8.0f}  max {d[:, i].max():8.0f}")
print("block start spread (cycles):", st[:, 0].max() - st[:, 0].min(), " end spread:", st[:, 5].max() - st[:, 5].min())
