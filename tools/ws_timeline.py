"""Developer aid: timeline of the three warp-specialised aggregate kernels (forward, backward by target "B1", backward by source "B2")
from six clock stamps per wave — wave entry, prologue done (behind the block's barrier), first tile published, loop end, last stores
issued, drained — none inside the steady loop.  Library built by `tools/build_prof_variant.sh tl` (-DGLAM_WS_TL in triplet_ws.hip and
triplet_ws_b1.hip), selected with GLAM_HIP_LIB.  Counters of different XCDs are not synchronised: every figure is relative to the start
of the block's own first wave; the launch's ramp / tail is the dispatch duration (glam_prof_*) minus the mean block lifetime.
usage: ws_timeline.py [batch]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from glam_amd import _lib, layer, ops
from glam_amd.data import synth_batch

dev = torch.device("cuda:0")
lib = _lib.load(); raw = ctypes.CDLL(_lib.LIB_PATH)
ops.USE_TORCH_EXT = False
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b = synth_batch(B, seed=0).to(dev)
torch.manual_seed(0)
conv = layer.TripletMessage(60, 4).to(dev)
x = torch.randn(b.x.size(0), 60, device=dev, requires_grad=True)
cot = torch.randn(b.x.size(0), 60, device=dev)
def step():
    out = conv(x, b.edge_index, b.edge_attr)
    torch.autograd.grad((out * cot).sum(), [x] + list(conv.parameters()))
for _ in range(5):
    step()
torch.cuda.synchronize()
with _lib.kernel_timer(capacity=512) as kt:
    for _ in range(10):
        step()
torch.cuda.synchronize()
dur = {}
for name, grid, us in kt.records():
    dur.setdefault(name, []).append(us)
dur = {k: sum(v) / len(v) for k, v in dur.items()}
N = b.x.size(0)
ntiles = (N + 15) // 16
buf = (ctypes.c_longlong * (2 * 256 * 12 * 6))()
assert raw.glam_debug_ws_tl(buf, len(buf)) == 0
tl = np.array(buf[:], dtype=np.int64).reshape(2, 256, 12, 6)
buf1 = (ctypes.c_longlong * (256 * 12 * 6))()
assert raw.glam_debug_b1_tl(buf1, len(buf1)) == 0
tl1 = np.array(buf1[:], dtype=np.int64).reshape(256, 12, 6)
bufr = (ctypes.c_longlong * (2 * 256 * 12 * 6))()
assert raw.glam_debug_ws_rt(bufr, len(bufr)) == 0
rt = np.array(bufr[:], dtype=np.int64).reshape(2, 256, 12, 6)
bufr1 = (ctypes.c_longlong * (256 * 12 * 6))()
assert raw.glam_debug_b1_rt(bufr1, len(bufr1)) == 0
rt1 = np.array(bufr1[:], dtype=np.int64).reshape(256, 12, 6)

def tl_valid(a):
    return a[:, 0, 0] > 0

def table(name, a, rt, nprod, label, vector_first):
    """a: [blocks, 12, 6]; the first `nprod` waves gather (or, B1: the first 8 are the vector waves), the rest run the matrix product"""
    a_all = a
    a = a[tl_valid(a)]
    nb = len(a)
    t0 = a[:, :, 0].min(axis=1)[:, None]
    g, m = a[:, :nprod], a[:, nprod:]
    life = (a[:, :, 5].max(axis=1) - t0[:, 0])
    us = dur.get(label)
    passes = ntiles * 4 / (nb * nprod)                 # 4-node passes per gathering wave
    print(f"{name}: {nb} blocks, {ntiles / nb:.2f} tiles per block, {passes:.2f} passes per gathering wave; dispatch {us:.2f} us" if us else name)
    rows = [("wave entry (spread inside the block)", a[:, :, 0] - t0),
            ("gather waves: prologue done (barrier passed)", g[:, :, 1] - t0),
            ("gather waves: first tile published / first pass done", g[:, :, 2] - t0),
            ("gather waves: loop end", g[:, :, 3] - t0),
            ("gather waves: last stores issued", g[:, :, 4] - t0),
            ("gather waves: drained", g[:, :, 5] - t0),
            ("matrix waves: loop end", m[:, :, 3] - t0),
            ("matrix waves: drained", m[:, :, 5] - t0)]
    if vector_first:      # B1: the matrix waves PRODUCE the tiles the vector waves wait for
        rows += [("matrix waves: barrier passed", m[:, :, 1] - t0), ("matrix waves: first tile published", m[:, :, 2] - t0),
                 ("matrix waves: fourth tile published (the ring is full)", m[:, :, 4] - t0)]
    for n_, v in rows:
        v = v[v > -1e15]
        print(f"   {n_:55s} mean {v.mean():8.0f}   min {v.min():8.0f}   max {v.max():8.0f}")
    first = (g[:, :, 2] - g[:, :, 1]).mean()
    steady = (g[:, :, 3] - g[:, :, 2]).mean() / max(passes - 1.0, 1e-9)
    print(f"   -> prologue {(g[:, :, 1] - t0).mean():.0f} | first pass exposed {first:.0f} | steady {steady:.0f} per further pass ({passes - 1:.2f} of them) | "
          f"drain {(a[:, :, 5].max(axis=1) - g[:, :, 3].max(axis=1)).mean():.0f} | block lifetime mean {life.mean():.0f}, max {life.max():.0f} cycles")
    # the launch as a whole on the device-wide 100 MHz counter (10 ns steps): when the blocks start and end relative to the launch's first stamp
    r = rt[tl_valid(a_all)]
    r0 = r[:, :, 0].min()
    b_start = (r[:, :, 0].min(axis=1) - r0) / 100.0
    b_end = (r[:, :, 5].max(axis=1) - r0) / 100.0
    g_pro = (r[:, :nprod, 1].max(axis=1) - r0) / 100.0
    span = b_end.max()
    print(f"   -> device-wide clock, us from the launch's first stamp: blocks start at mean {b_start.mean():.2f} (90 % by {np.quantile(b_start, 0.9):.2f}, last {b_start.max():.2f}), "
          f"pass their barrier at mean {g_pro.mean():.2f}, end at mean {b_end.mean():.2f} (10 % by {np.quantile(b_end, 0.1):.2f}, last {span:.2f}); "
          f"block lifetime mean {(b_end - b_start).mean():.2f} us; CU-time idle before the first and after the last wave of a block: "
          f"{b_start.mean():.2f} + {(span - b_end).mean():.2f} us of {span:.2f}")
    if us:
        print(f"   -> at the dispatch's {us:.2f} us a mean block lifetime of {life.mean():.0f} cycles leaves {us - life.mean() / 2100:.2f} us of ramp + tail at 2.1 GHz "
              f"({us - life.mean() / 1900:.2f} at 1.9)")

print(f"B = {B}: N = {N}, {ntiles} tiles; clock cycles relative to the start of the block's first wave")
table("forward  k_triplet_fwd_ws", tl[0], rt[0], 8, "k_triplet_fwd_ws+update", False)
table("backward by target  k_triplet_bwd_dst_ws (B1; gather = its 8 vector waves, matrix = the 4 d_aggr waves)", tl1, rt1, 8, "d_aggr+k_triplet_bwd_dst_ws", True)
table("backward by source  k_triplet_bwd_src_ws (B2)", tl[1], rt[1], 8, "k_triplet_bwd_src_ws+dx", False)
