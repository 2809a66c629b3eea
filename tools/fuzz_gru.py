"""Randomised sweep of the warp-specialised GRU launches (glam_gru_ws_fwd, glam_gru_bwd_ws, and both on the pre-split images) against the fp64 gate equations of
torch.nn.GRU and their autograd: widths 24..64, ragged row counts, with / without residual, folded CELU, hidden-state gradient,
merged identity.  usage: fuzz_gru.py [n] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from glam_amd import _lib
dev = torch.device("cuda")
lib, p = _lib.load(), _lib.ptr
st = lambda: torch.cuda.current_stream().cuda_stream
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    torch.manual_seed(1000 * (int(sys.argv[2]) if len(sys.argv) > 2 else 0) + case)
    C = int(rng.integers(6, 17)) * 4
    N = int(rng.choice([1, 15, 16, 17, 31, 100, 1000, 4097, 20400]))
    celu, ident, hstate = bool(rng.random() < 0.5), bool(rng.random() < 0.7), bool(rng.random() < 0.6)
    merge = ident and bool(rng.random() < 0.3)
    act = int(rng.integers(0, 4))                      # none, relu, leaky, celu
    slope = 0.1
    M = 3 * C
    r = lambda *s: torch.randn(*s, dtype=torch.float64)
    x, h, idn, w_ih, w_hh, b_ih, b_hh = r(N, C), r(N, C), r(N, C), r(M, C) * 0.3, r(M, C) * 0.3, r(M), r(M)
    if merge:
        idn = h
    xs, hs = x.clone().requires_grad_(True), h.clone().requires_grad_(True)
    ids = hs if merge else idn.clone().requires_grad_(True)
    xin = torch.nn.functional.celu(xs) if celu else xs
    gi_r, gh_r = xin @ w_ih.t() + b_ih, hs @ w_hh.t() + b_hh
    rr, zz = torch.sigmoid(gi_r[:, :C] + gh_r[:, :C]), torch.sigmoid(gi_r[:, C:2 * C] + gh_r[:, C:2 * C])
    nn_ = torch.tanh(gi_r[:, 2 * C:] + rr * gh_r[:, 2 * C:])
    hn_r = (1 - zz) * nn_ + zz * hs
    y = hn_r + (ids if ident else 0)
    out_r = [y, torch.relu(y), torch.nn.functional.leaky_relu(y, slope), torch.nn.functional.celu(y)][act]
    d_out, d_hs = r(N, C), r(N, C)
    if act in (1, 2):      # no gradient through elements at the kink: fp32 and fp64 may land on different sides of it (1 in ~1e6 elements)
        d_out = d_out * (y.detach().abs() > 1e-5)
    loss = (out_r * d_out).sum() + ((hn_r * d_hs).sum() if hstate else 0)
    grads = torch.autograd.grad(loss, [xs, hs] + ([] if (merge or not ident) else [ids]))
    f32 = lambda t: t.detach().float().to(dev).contiguous()
    xd, hd, idd, wi, wh, bi, bh = (f32(t) for t in (x, h, idn, w_ih, w_hh, b_ih, b_hh))
    ia, ib = (torch.empty(lib.glam_ts_gemm_image_bytes(C, M) // 4, device=dev) for _ in range(2))
    ta, tb = (torch.empty(lib.glam_ts_gemm_image_bytes(M, C) // 4, device=dev) for _ in range(2))
    for w, i, t in ((wi, ia, ta), (wh, ib, tb)):
        assert lib.glam_ts_gemm_make_image(p(w), C, 1, C, M, p(i), st()) == 0
        assert lib.glam_ts_gemm_make_image(p(w), C, 0, M, C, p(t), st()) == 0
    nan = lambda *s: torch.full(s, float("nan"), device=dev)
    gi, gh, hn, out = nan(N, M), nan(N, M), nan(N, C), nan(N, C)
    idp = hd if merge else idd
    try:
        rc = lib.glam_gru_ws_fwd(p(xd), p(hd), p(idp) if ident else None, p(ia), p(ib), p(bi), p(bh), N, C, int(celu), act, slope, p(gi), p(gh), p(hn), p(out), st())
        assert rc == 0, lib.glam_last_error()
        tol = lambda a, b: (a.double().cpu() - b.detach()).abs().max().item() / max(1.0, b.detach().abs().max().item())
        e = [tol(gi, gi_r), tol(gh, gh_r), tol(hn, hn_r), tol(out, out_r)]
        assert max(e) < 3e-6, f"forward {e}"
        dgi, dgh, did, dx, dh = nan(N, M), nan(N, M), nan(N, C), nan(N, C), nan(N, C)
        d_out_d, d_hs_d = f32(d_out), f32(d_hs)        # (kept alive across the call: a temporary's memory would be handed to the next one)
        rc = lib.glam_gru_bwd_ws(p(gi), p(gh), p(hd), p(out), p(d_out_d), p(d_hs_d) if hstate else None, p(xd), p(ta), p(tb), N, C, int(celu), act, slope,
                                 int(merge), p(dgi), p(dgh), p(did) if ident else None, p(dx), p(dh), st())
        assert rc == 0, lib.glam_last_error()
        e = [tol(dx, grads[0]), tol(dh, grads[1])] + ([tol(did, grads[2])] if (ident and not merge) else [])
        assert max(e) < 1e-5, f"backward {e}"
        # the same launches on the pre-split images of the gate matrices (glam_gru_ws_make_pre): every output bit for bit
        pre = torch.empty(2, lib.glam_gru_ws_pre_bytes(), dtype=torch.uint8, device=dev)
        assert lib.glam_gru_ws_make_pre(p(wi), p(wh), C, p(pre[0]), p(pre[1]), st()) == 0, lib.glam_last_error()
        gi2, gh2, hn2, out2 = nan(N, M), nan(N, M), nan(N, C), nan(N, C)
        rc = lib.glam_gru_ws_fwd_pre(p(xd), p(hd), p(idp) if ident else None, p(pre[0]), p(bi), p(bh), N, C, int(celu), act, slope, p(gi2), p(gh2), p(hn2),
                                     p(out2), None, st())
        assert rc == 0, lib.glam_last_error()
        assert all(torch.equal(u, v) for u, v in ((gi, gi2), (gh, gh2), (hn, hn2), (out, out2))), "forward on the pre-split image differs"
        o2 = [nan(N, M), nan(N, M), nan(N, C), nan(N, C), nan(N, C)]
        rc = lib.glam_gru_bwd_ws_pre(p(gi), p(gh), p(hd), p(out), p(d_out_d), p(d_hs_d) if hstate else None, p(xd), p(pre[1]), N, C, int(celu), act, slope,
                                     int(merge), p(o2[0]), p(o2[1]), p(o2[2]) if ident else None, p(o2[3]), p(o2[4]), st())
        assert rc == 0, lib.glam_last_error()
        pairs = [(dgi, o2[0]), (dgh, o2[1]), (dx, o2[3]), (dh, o2[4])] + ([(did, o2[2])] if (ident and not merge) else [])
        assert all(torch.equal(u, v) for u, v in pairs), "backward on the pre-split image differs"
    except AssertionError as ex:
        bad += 1
        print(f"case {case}: N={N} C={C} celu={celu} ident={ident} hstate={hstate} merge={merge} act={act}: {ex}")
print(f"{n_cases - bad}/{n_cases} cases ok")
sys.exit(1 if bad else 0)
