"""Developer aid: cycle stamps of the fused forward GRU step (library built by tools/build_prof_variant.sh gru, GLAM_HIP_LIB=...)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from glam_amd import _lib, layer, ops
from glam_amd.data import synth_batch

ops.GRU_FUSED = "1"
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b = synth_batch(B, seed=0).to(dev)
blk = layer.MessageBlock(60, 60, 4, norm="_None", dropout="_None()", conv="_TripletMessage", act="ReLU", res=True).to(dev).eval()
x = torch.randn(b.x.size(0), 60, device=dev)
with torch.no_grad():
    for _ in range(3):
        blk(x, b.edge_index, b.edge_attr, h=None, batch=b.batch)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * (256 * 8))()
assert raw.glam_debug_gru_prof(buf, 256 * 8) == 0
st = np.array(buf[:], dtype=np.int64).reshape(256, 8)
names = ["A loads issued, images staged, barrier", "MFMAs of the first tile (wave 0)", "epilogue of the first tile", "later tiles", "drain stores"]
for i, n in enumerate(names):
    d = st[:, i + 1] - st[:, i]
    print(f"  {n:42s} mean {d.mean():8.0f}  max {d.max():8.0f}")
print("  block lifetime mean", (st[:, 5] - st[:, 0]).mean(), " span", st[:, 5].max() - st[:, 0].min())
