"""Developer aid: cycle breakdown of a k_triplet_bwd_dst wave (library built with -DGLAM_B1_PROF in triplet_h3.hip,
pointed to by GLAM_HIP_LIB).  The profiling build drains the memory queues at every stamp, so the phases are
serialised: it shows where the latency is, not the overlapped total."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from glam_amd import _lib, layer
from glam_amd.data import synth_batch

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b = synth_batch(B, seed=0).to(dev)
conv = layer.TripletMessage(60, 4).to(dev)
x = torch.randn(b.x.size(0), 60, device=dev)
print(bench.time_kernels(conv, b, x, reps=20))
raw = ctypes.CDLL(_lib.LIB_PATH)
nb = 512
buf = (ctypes.c_longlong * (nb * 8))()
assert raw.glam_debug_b1_prof(buf, nb * 8) == 0
st = np.array(buf[:], dtype=np.int64).reshape(nb, 8)
names = ["rowptr/a_i/stats", "d_aggr,aggr rows + dot", "indices", "neighbour rows/ea/a_j", "compute", "stores"]
for i, n in enumerate(names):
    print(f"  {n:26s} mean {st[:, i].mean():8.0f}  max {st[:, i].max():8.0f}")
print("  wave total (loop)         mean", st[:, :6].sum(1).mean(), " epilogue+loop wall:", (st[:, 7] - st[:, 6]).mean(),
      " start spread:", st[:, 6].max() - st[:, 6].min(), " end spread:", st[:, 7].max() - st[:, 7].min(),
      " kernel span:", st[:, 7].max() - st[:, 6].min())
