"""Developer aid: where a k_triplet_fwd_dma wave spends its cycles (library built with tools/build_prof_variant.sh dma, GLAM_HIP_LIB=...)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from glam_amd import _lib, layer, ops
from glam_amd.data import synth_batch
dev = torch.device("cuda:0")
lib = _lib.load(); raw = ctypes.CDLL(_lib.LIB_PATH)
p, st = _lib.ptr, _lib.stream
torch.manual_seed(0)
conv = layer.TripletMessage(60, 4).to(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
b = synth_batch(B, seed=7).to(dev)
N, E = b.x.size(0), b.edge_index.size(1)
x = torch.randn(N, 60, device=dev)
gi = ops.graph_index(b.edge_index, N)
with torch.no_grad():
    Wn, Wa, We, M, Ws, Cp, Dp = conv._staged_weights()
    xw, a_ij = (x @ Wn).contiguous(), (x @ Wa).contiguous()
aggr, stats = torch.empty(N, 180, device=dev), torch.empty(N, 8, device=dev)
es, ee = torch.empty(N, 4, dtype=torch.int32, device=dev), torch.empty(N, 4, dtype=torch.int32, device=dev)
ovf = torch.zeros(1, dtype=torch.int32, device=dev)
lib.glam_ell_build(p(gi.rowptr), p(gi.src), p(gi.eid), N, p(es), p(ee), p(ovf), st())
for _ in range(3):
    lib.glam_triplet_fwd_ell(p(xw), p(a_ij), p(b.edge_attr), p(We), p(M), p(es), p(ee), N, E, 3, Cp, Dp, 0.2, 1, p(aggr), p(stats), 512, st())
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 512)()
assert raw.glam_debug_dma_prof(buf, 512) == 0
a = np.array(buf[:], dtype=np.int64).reshape(64, 8)
npass = (N + 3) // 4 / 2048
names = ["loop/prev-compute tail", "wait vmcnt(0)", "stores", "dma / row-load issue", "rec load", "compute", "(pipe: whole second half-trip)"]
if os.environ.get("GLAM_ELL_VARIANT") != "dma":
    npass /= 2      # the register-pipelined kernel stamps the first of the two passes of a loop trip
print(f"B={B}: passes per wave {npass:.1f}; cycles per pass (mean over 64 waves):")
for k, n in enumerate(names):
    print(f"   {n:24s} {a[:, k].mean() / npass:9.0f}")
print("   total", a[:, :7].sum(1).mean() / npass)
