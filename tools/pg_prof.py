"""Developer aid: cycle stamps of k_param_grads per block role (library built by tools/build_prof_variant.sh pg, GLAM_HIP_LIB=...)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from glam_amd import _lib, layer
from glam_amd.data import synth_batch

dev = torch.device("cuda:0")
b = synth_batch(1024, seed=0).to(dev)
conv = layer.TripletMessage(60, 4).to(dev)
x = torch.randn(b.x.size(0), 60, device=dev, requires_grad=True)
for _ in range(5):
    out = conv(x, b.edge_index, b.edge_attr)
    out.sum().backward()
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * (512 * 8))()
assert raw.glam_debug_pg_prof(buf, 512 * 8) == 0
st = np.array(buf[:], dtype=np.int64).reshape(512, 8)
nA, nB, nC = 48, 48, 3
st = st[st[:, 0] != 0]
t0 = st[:, 0].min()
roles = {"A": st[:nA], "B": st[nA:nA + nB], "C": st[nA + nB:nA + nB + nC], "D": st[nA + nB + nC:]}
print("cycles relative to the first block's entry (readcyclecounter); blocks", len(st))
for name, r in roles.items():
    if len(r) == 0:
        continue
    line = f"  {name}: entry {np.mean(r[:, 0] - t0):8.0f} (max {np.max(r[:, 0] - t0):6.0f})   end {np.mean(r[:, 4] - t0):8.0f} (max {np.max(r[:, 4] - t0):6.0f})"
    if name == "D":
        line += f"   sum1 {np.mean(r[:, 2] - r[:, 1]):7.0f}  sum2 {np.mean(r[:, 3] - r[:, 2]):7.0f}  tail {np.mean(r[:, 4] - r[:, 3]):7.0f}  pre {np.mean(r[:, 1] - r[:, 0]):6.0f}"
    print(line)
