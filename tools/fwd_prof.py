"""Developer aid: cycle breakdown of the fused forward kernel (aggregate + update epilogue) per block (library built by
tools/build_prof_variant.sh fwd, GLAM_HIP_LIB=...)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from glam_amd import _lib, layer
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
buf = (ctypes.c_longlong * (512 * 8))()
assert raw.glam_debug_fwd_prof(buf, 512 * 8) == 0
st = np.array(buf[:], dtype=np.int64).reshape(512, 8)
st = st[st[:, 7] != 0]
names = ["loop top", "edge phase (gather, softmax, aggregate, aggr / stats stores, tile to LDS)", "barrier 1 (__syncthreads: vmcnt + lgkmcnt)",
         "48 MFMAs + s_out writes", "barrier 2", "out tile: LDS read, bias, store"]
for i, n in enumerate(names):
    print(f"  {n:76s} mean {st[:, i].mean():8.0f}  max {st[:, i].max():8.0f}")
print("  block lifetime (after start-up) mean", (st[:, 7] - st[:, 6]).mean(), " blocks", len(st))
