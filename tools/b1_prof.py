"""Developer aid: cycle breakdown of a k_triplet_bwd_dst wave (library built by tools/build_prof_variant.sh b1 | b1n,
pointed to by GLAM_HIP_LIB).  The b1 build drains the memory queues at every stamp, so the phases are serialised: it shows
where the latency is; b1n does not drain: the overlapped picture."""
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
x = torch.randn(b.x.size(0), 60, device=dev, requires_grad=True)
for _ in range(5):
    out = conv(x, b.edge_index, b.edge_attr)
    out.backward(torch.randn_like(out))
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
nb = 512
buf = (ctypes.c_longlong * (nb * 32))()
assert raw.glam_debug_b1_prof(buf, nb * 32) == 0
st = np.array(buf[:], dtype=np.int64).reshape(nb, 32)
st = st[st[:, 7] != 0]
names = {0: "rowptr/a_i/stats (+MFMA-phase tail)", 1: "d_aggr,aggr rows + dot", 2: "indices", 3: "neighbour rows/ea/a_j", 4: "compute",
         5: "stores", 8: "loop top", 9: "wait vmcnt(0) (A tile, stores)", 10: "barrier 0", 16: "node loads issue", 17: "A fragment reads",
         18: "B reads + MFMAs", 11: "tile writes",
         12: "barrier 1", 13: "dag reads, barrier 2, next A fetch"}
for i, n in names.items():
    print(f"  {n:46s} mean {st[:, i].mean():8.0f}  max {st[:, i].max():8.0f}")
print("  start-up (kernel entry -> loop)   mean", (st[:, 6] - st[:, 14]).mean(), " loop+epilogue wall:", (st[:, 7] - st[:, 6]).mean(),
      " block lifetime:", (st[:, 7] - st[:, 14]).mean(), " kernel span:", st[:, 7].max() - st[:, 14].min())
