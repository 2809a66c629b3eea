"""Developer aid: where the producer / consumer waves of k_triplet_fwd_ws spend their cycles (library built with
tools/build_prof_variant.sh ws, run with GLAM_HIP_LIB=tools/tmp/variants/lib_wsprof.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from glam_amd import _lib, layer, ops
from glam_amd.data import synth_batch
dev = torch.device("cuda:0")
lib = _lib.load(); raw = ctypes.CDLL(_lib.LIB_PATH)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
b = synth_batch(B, seed=7).to(dev)
torch.manual_seed(0)
conv = layer.TripletMessage(60, 4).to(dev)
x = torch.randn(b.x.size(0), 60, device=dev)
with torch.no_grad():
    for _ in range(3):
        conv(x, b.edge_index, b.edge_attr)
torch.cuda.synchronize()
P = 8
buf = (ctypes.c_longlong * 6144)()
assert raw.glam_debug_ws_prof(buf, 6144) == 0
a = np.array(buf[:], dtype=np.int64).reshape(64, 12, 8)
N = b.x.size(0)
grid = min((N + 15) // 16, int(os.environ.get("GLAM_WS_GRID", "256")))
tiles = (N + 15) // 16 / grid
print(f"B={B}: {tiles:.1f} tiles per block; cycles per tile (mean over 64 blocks)")
pn = ["loop top", "wait vmcnt(0)", "stores", "(unused)", "ballot / DPP / side-table DMA + row loads issue", "record load", "compute", "publish (incl. wait for a free slot)"]
print(f"  producer waves 0-{P - 1} (cycles per PASS: each wave gathers every {P // 4}-th tile of the block):")
tiles_p = tiles / (P // 4)
for k, n in enumerate(pn):
    print(f"     {n:40s} {a[:, :P, k].mean() / tiles_p:9.0f}")
print(f"     {'total':40s} {a[:, :P, :8].sum(2).mean() / tiles_p:9.0f}")
cn = ["wait for a published tile", "A fragments (12 ds_read_b128) + check-out", "48 MFMAs + out stores"]
print(f"  consumer waves {P}-{P + 3} (cycles per tile):")
for k, n in enumerate(cn):
    print(f"     {n:40s} {a[:, P:P + 4, k].mean() / tiles:9.0f}")
print(f"     {'total':40s} {a[:, P:P + 4, :3].sum(2).mean() / tiles:9.0f}")
