"""Developer aid: clock stamps of k_wgrad_x3's producer and consumer waves (library built with -DGLAM_WX_PROF by
tools/build_prof_variant.sh wx, pointed to by GLAM_HIP_LIB): where a launch of the headline step's two weight-gradient products spends
its time — launch ramp, first rows in LDS, the streaming loop, the partial stores.  usage: wx_prof.py [N]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from glam_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20400
p, st = _lib.ptr, _lib.stream
Pa, Qa = torch.randn(N, 180, device=dev), torch.randn(N, 60, device=dev)
Pb, Qb = torch.randn(N, 188, device=dev), torch.randn(N, 60, device=dev)
oa, ob = torch.empty(181, 60, device=dev), torch.empty(188, 60, device=dev)
ws = torch.empty(lib.glam_wgrad_workspace_bytes(), dtype=torch.uint8, device=dev)
def run():
    _lib.check(lib.glam_wgrad_gemm_pair(p(Pa), 180, 180, 1, p(Qa), 60, 60, 0, 0, p(oa), 60, 1, p(Pb), 188, 188, 0, p(Qb), 60, 60, 0, 0, p(ob), 60, 1,
                                        N, p(ws), ws.numel(), st()), "pair")
for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
print(f"N={N}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (product launch + k_final_reduce, eager)")
nb = 512
buf = (ctypes.c_longlong * (nb * 16 * 8))()
assert raw.glam_debug_wx_prof(buf, nb * 16 * 8) == 0
a = np.array(buf[:], dtype=np.int64).reshape(nb, 16, 8)
a = a[a[:, 8, 3] > 0]                       # blocks that ran (a consumer's last stamp)
t0 = a[:, :, 0].min(axis=1)[:, None]        # per block: the counters of different XCDs are not synchronised
prod, cons = a[:, :8], a[:, 8:]
print(f"blocks {len(a)}; stamps in clock ticks relative to the start of the block's first wave")
def row(name, v):
    print(f"  {name:58s} mean {v.mean():9.0f}  min {v.min():9.0f}  max {v.max():9.0f}")
row("wave start", a[:, :, 0] - t0)
row("arguments decoded", a[:, :, 5] - t0)
row("behind the block barrier", a[:, :, 6] - t0)
row("producer: first loads issued", prod[:, :, 1] - t0)
row("producer: first step split + written", prod[:, :, 3] - t0)
row("consumer: first stage ready", cons[:, :, 4] - t0)
row("producer: loop done", prod[:, :, 2] - t0)
row("consumer: loop done", cons[:, :, 2] - t0)
row("consumer: partial stored", cons[:, :, 3] - t0)
row("producer: loop length", prod[:, :, 2] - prod[:, :, 1])
row("consumer: loop length (from first ready)", cons[:, :, 2] - cons[:, :, 4])
row("producer: waiting for its rows (sum over its steps)", prod[:, :, 4])
row("producer: waiting for the ring slot (sum)", prod[:, :, 7])
row("consumer: waiting for a stage (sum, first excluded)", cons[:, :, 1])
row("consumer: fragment reads issue -> arrival (sum)", cons[:, :, 7])
row("block lifetime", (a[:, 8:, 3] - t0).max(axis=1))
