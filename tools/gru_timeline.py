"""Developer aid: timeline of the two warp-specialised GRU kernels (k_gru_fwd_ws: 4 producer + 4 consumer waves; k_gru_bwd_ws: 4 + 8) from
stamps per wave — entry, prologue done (producers: first loads issued; consumers: weight slice split), first tile done, loop end, drained —
on the CU's shader clock and on the device-wide 100 MHz counter.  Library built by `tools/build_prof_variant.sh tl`, selected with
GLAM_HIP_LIB.  usage: gru_timeline.py [rows]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from glam_amd import _lib
lib, p = _lib.load(), _lib.ptr
raw = ctypes.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda")
st = lambda: torch.cuda.current_stream().cuda_stream
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20400
C, M = 60, 180
r = lambda *s: torch.randn(*s, device=dev)
x, h, idn, w_ih, w_hh, b_ih, b_hh = r(N, C), r(N, C), r(N, C), r(M, C) * 0.3, r(M, C) * 0.3, r(M), r(M)
ia, ib = (torch.empty(lib.glam_ts_gemm_image_bytes(C, M) // 4, device=dev) for _ in range(2))
ta, tb = (torch.empty(lib.glam_ts_gemm_image_bytes(M, C) // 4, device=dev) for _ in range(2))
for w, i, t in ((w_ih, ia, ta), (w_hh, ib, tb)):
    assert lib.glam_ts_gemm_make_image(p(w), C, 1, C, M, p(i), st()) == 0
    assert lib.glam_ts_gemm_make_image(p(w), C, 0, M, C, p(t), st()) == 0
gi, gh, hn, out = torch.empty(N, M, device=dev), torch.empty(N, M, device=dev), torch.empty(N, C, device=dev), torch.empty(N, C, device=dev)
d_out, d_hs = r(N, C), r(N, C)
dgi, dgh, did, dx, dh = torch.empty(N, M, device=dev), torch.empty(N, M, device=dev), torch.empty(N, C, device=dev), torch.empty(N, C, device=dev), torch.empty(N, C, device=dev)
PRE = os.environ.get("PRE", "1") != "0"          # the pre-split images (the model's route) or the plain k_ts_gemm images
pre = torch.empty(2, lib.glam_gru_ws_pre_bytes(), dtype=torch.uint8, device=dev)
assert lib.glam_gru_ws_make_pre(p(w_ih), p(w_hh), C, p(pre[0]), p(pre[1]), st()) == 0
def fwd_pre(): assert lib.glam_gru_ws_fwd_pre(p(x), p(h), p(idn), p(pre[0]), p(b_ih), p(b_hh), N, C, 1, 1, 0.0, p(gi), p(gh), p(hn), p(out), None, st()) == 0
def bwd_pre(): assert lib.glam_gru_bwd_ws_pre(p(gi), p(gh), p(h), p(out), p(d_out), p(d_hs), p(x), p(pre[1]), N, C, 1, 1, 0.0, 0, p(dgi), p(dgh), p(did), p(dx), p(dh), st()) == 0
def fwd(): assert lib.glam_gru_ws_fwd(p(x), p(h), p(idn), p(ia), p(ib), p(b_ih), p(b_hh), N, C, 1, 1, 0.0, p(gi), p(gh), p(hn), p(out), st()) == 0
def bwd(): assert lib.glam_gru_bwd_ws(p(gi), p(gh), p(h), p(out), p(d_out), p(d_hs), p(x), p(ta), p(tb), N, C, 1, 1, 0.0, 0, p(dgi), p(dgh), p(did), p(dx), p(dh), st()) == 0
if PRE: fwd, bwd = fwd_pre, bwd_pre
if os.environ.get("NODE", "0") != "0" or os.environ.get("GATES", "0") != "0":
    # the model's forward: the gates kept (gh = NULL) and, with NODE=1, the node product of the block's next application inside the launch
    H = 3
    G4 = torch.empty(N, 4 * C, device=dev)
    wn, we, att, wsc, bias = r(C, H * C) * 0.2, r(4, H * C) * 0.2, r(1, H, 3 * C) * 0.2, r(H * C, C) * 0.2, r(C)
    staged = torch.empty(lib.glam_triplet_staged_floats(H, C, 4), device=dev)
    assert lib.glam_triplet_stage_params(p(wn), p(we), p(att), p(wsc), p(bias), C, H, 4, C, 4, p(staged), st()) == 0
    nimg = staged[lib.glam_triplet_staged_node_fragments(H, C, 4):]
    xw, a_ij, xc = torch.empty(N, H * C, device=dev), torch.empty(N, 8, device=dev), torch.empty(N, C, device=dev)
    D4 = torch.empty(N, 4 * C, device=dev)
    def bwd(): assert lib.glam_gru_bwd_ws_pre(p(G4), None, p(h), p(out), p(d_out), p(d_hs), p(xc), p(pre[1]), N, C, 2, 1, 0.0, 0, p(D4), None, p(did), p(dx), p(dh), st()) == 0
    if os.environ.get("NODE", "0") != "0":
        def fwd(): assert lib.glam_gru_ws_fwd_pre_node(p(x), p(h), p(idn), p(pre[0]), p(b_ih), p(b_hh), N, C, 1, 1, 0.0, p(G4), None, p(hn), p(out), p(xc), p(nimg), H * C, p(xw), p(a_ij), st()) == 0
    else:
        def fwd(): assert lib.glam_gru_ws_fwd_pre(p(x), p(h), p(idn), p(pre[0]), p(b_ih), p(b_hh), N, C, 1, 1, 0.0, p(G4), None, p(hn), p(out), p(xc), st()) == 0
for _ in range(5): fwd(); bwd()
torch.cuda.synchronize()
with _lib.kernel_timer(capacity=64) as kt:
    for _ in range(5): fwd(); bwd()
torch.cuda.synchronize()
dur = {}
for name, grid, us in kt.records():
    dur.setdefault(name, []).append(us)
print({k: round(sum(v) / len(v), 2) for k, v in dur.items()})
n = 2 * 256 * 12 * 6
def get(device_wide):
    buf = (ctypes.c_longlong * n)()
    assert raw.glam_debug_gru_tl(buf, n, device_wide) == 0
    return np.array(buf[:], dtype=np.int64).reshape(2, 256, 12, 6)
tl, rt = get(0), get(1)
ntiles = (N + 15) // 16
for kid, name, nwaves in ((0, "k_gru_fwd_ws (4 producers + 4 consumers)", 8), (1, "k_gru_bwd_ws (4 producers + 8 consumers)", 12)):
    a, rr = tl[kid][:, :nwaves], rt[kid][:, :nwaves]
    ok = a[:, 0, 0] > 0
    a, rr = a[ok], rr[ok]
    t0 = a[:, :, 0].min(axis=1)[:, None]
    g, m = a[:, :4], a[:, 4:]
    print(f"{name}: {len(a)} blocks, {ntiles / len(a):.2f} tiles per block, N = {N}")
    if kid == 0:
        for label, v in (("producers: barrier passed", g[:, :, 4] - t0), ("consumers: weight loads issued", m[:, :, 1] - t0), ("consumers: barrier passed", m[:, :, 4] - t0)):
            print(f"   {label:36s} mean {v.mean():8.0f}   min {v.min():8.0f}   max {v.max():8.0f}")
    for label, v in (("wave entry", a[:, :, 0] - t0), ("producers: first loads issued", g[:, :, 1] - t0), ("producers: first tile published", g[:, :, 2] - t0),
                     ("producers: loop end", g[:, :, 3] - t0), ("producers: drained", g[:, :, 5] - t0), ("consumers: weight slice split", m[:, :, 1] - t0),
                     ("consumers: first tile done", m[:, :, 2] - t0), ("consumers: loop end", m[:, :, 3] - t0), ("consumers: drained", m[:, :, 5] - t0)):
        print(f"   {label:36s} mean {v.mean():8.0f}   min {v.min():8.0f}   max {v.max():8.0f}")
    life = a[:, :, 5].max(axis=1) - t0[:, 0]
    r0 = rr[:, :, 0].min()
    bs, be = (rr[:, :, 0].min(axis=1) - r0) / 100.0, (rr[:, :, 5].max(axis=1) - r0) / 100.0
    print(f"   block lifetime mean {life.mean():.0f} cycles, max {life.max():.0f}; device-wide clock: blocks start at mean {bs.mean():.2f} us (last {bs.max():.2f}), "
          f"end at mean {be.mean():.2f} (last {be.max():.2f})")

# phases inside a consumer tile of the forward: the first and the third of the wave
buf = (ctypes.c_longlong * (256 * 12 * 16))()
if hasattr(raw, "glam_debug_gru_finef") and raw.glam_debug_gru_finef(buf, len(buf)) == 0:
    fz = np.array(buf[:], dtype=np.int64).reshape(256, 12, 16)[:, 4:8]
    names = ["wait for the tile", "LDS reads + 72 matrix instructions", "bias + 6 stores (gi, gh)", "gates + 2 stores"]
    for base, what in ((0, "first"), (8, "third")):
        print(f"k_gru_fwd_ws consumers, the wave's {what} tile:")
        for k, nme in enumerate(names):
            a0, a1 = fz[:, :, base + k], fz[:, :, base + k + 1]
            d = (a1 - a0)[(a0 > 0) & (a1 > 0)]
            if d.size: print(f"   {nme:48s} mean {d.mean():7.0f}   min {d.min():7.0f}   max {d.max():7.0f}")
    print("k_gru_fwd_ws consumers, before the first tile (cycles between stamps):")
    for nme, k0, k1 in (("wait for the producers' check-in", 5, 6), ("24 weight loads issued", 6, 7), ("... arrived", 7, 13), ("split into 36 fragments (+ bias loads)", 13, 0)):
        a0, a1 = fz[:, :, k0], fz[:, :, k1]
        d = (a1 - a0)[(a0 > 0) & (a1 > 0)]
        if d.size: print(f"   {nme:48s} mean {d.mean():7.0f}   min {d.min():7.0f}   max {d.max():7.0f}")
