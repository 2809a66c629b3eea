"""Randomised sweep of the launches that write a TripletMessage's node product with the rows themselves (glam_gru_ws_(rng_)fwd_pre_node,
glam_ts_gemm_act_node; csrc/node_product.h) and of the GRU step on its gates (gh = NULL): widths 24..64, 1..4 heads, ragged and tiny row
counts, with / without residual, folded CELU, RReLU / Dropout — every output against the stand-alone launches, bit for bit.
usage: fuzz_node.py [n] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from glam_amd import _lib, ops
dev = torch.device("cuda")
lib, p = _lib.load(), _lib.ptr
st = lambda: torch.cuda.current_stream().cuda_stream
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
f = lambda *s: torch.full(s, float("nan"), device=dev)
bad = 0
for case in range(n_cases):
    torch.manual_seed(1000 * seed + case)
    C = int(rng.integers(6, 17)) * 4
    hs = [H for H in (1, 2, 3, 4) if 56 < H * C <= 184]
    if not hs:
        continue
    H = int(rng.choice(hs))
    N = int(rng.choice([1, 15, 16, 17, 31, 33, 100, 1000, 4097, 20400]))
    celu, ident, train = bool(rng.random() < 0.5), bool(rng.random() < 0.7), bool(rng.random() < 0.5)
    dp = float(rng.choice([0.0, 0.2, 0.5])) if train else 0.0
    M, HC = 3 * C, H * C
    r = lambda *s: torch.randn(*s, device=dev)
    x, h, idn, w_ih, w_hh, b_ih, b_hh = r(N, C), r(N, C), r(N, C), r(M, C) * 0.3, r(M, C) * 0.3, r(M), r(M)
    wn, we, att, wsc, bias = r(C, HC) * 0.2, r(4, HC) * 0.2, r(1, H, 3 * C) * 0.2, r(HC, C) * 0.2, r(C)
    staged = torch.empty(lib.glam_triplet_staged_floats(H, C, 4), device=dev)
    assert lib.glam_triplet_stage_params(p(wn), p(we), p(att), p(wsc), p(bias), C, H, 4, C, 4, p(staged), st()) == 0
    nimg, nfrag = staged[lib.glam_triplet_staged_node_image(H, C, 4):], staged[lib.glam_triplet_staged_node_fragments(H, C, 4):]
    pre = torch.empty(2, lib.glam_gru_ws_pre_bytes(), dtype=torch.uint8, device=dev)
    assert lib.glam_gru_ws_make_pre(p(w_ih), p(w_hh), C, p(pre[0]), p(pre[1]), st()) == 0
    lo, hi = 0.125, 1.0 / 3
    act = 4 if train else int(rng.integers(0, 4))
    res = []
    for node in (False, True):
        G, hn, out, drop, xc, xw, a_ij = f(N, 4 * C), f(N, C), f(N, C), f(N, C), f(N, C), f(N, HC), f(N, 8)
        state = torch.tensor([91 + case] + [0] * (ops.RNG_STATE_WORDS - 1), dtype=torch.int64, device=dev)
        eff = torch.zeros(2, dtype=torch.int64, device=dev)
        idp, xcp = (p(idn) if ident else None), (p(xc) if celu else None)
        tail = (p(nfrag), HC, p(xw), p(a_ij), st()) if node else (st(),)
        if train:
            fn = lib.glam_gru_ws_rng_fwd_pre_node if node else lib.glam_gru_ws_rng_fwd_pre
            rc = fn(p(x), p(h), idp, p(pre[0]), p(b_ih), p(b_hh), N, C, int(celu), act, 0.1, lo, hi, dp, p(state), p(eff), p(G), None, p(hn), p(out),
                    p(drop) if dp > 0 else None, xcp, *tail)
        else:
            fn = lib.glam_gru_ws_fwd_pre_node if node else lib.glam_gru_ws_fwd_pre
            rc = fn(p(x), p(h), idp, p(pre[0]), p(b_ih), p(b_hh), N, C, int(celu), act, 0.1, p(G), None, p(hn), p(out), xcp, *tail)
        assert rc == 0, lib.glam_last_error()
        res.append((G, hn, out, drop, xc, xw, a_ij))
    ok = all(torch.equal(u, v) or (torch.isnan(u).all() and torch.isnan(v).all()) for u, v in zip(res[0][:5], res[1][:5]))
    rows = res[1][3] if (train and dp > 0) else res[1][2]
    want_xw, want_a = f(N, HC), f(N, 8)
    assert lib.glam_ts_gemm(p(rows), C, C, None, 0, 0, p(nimg), None, p(want_xw), HC, HC, p(want_a), 8, 8, N, st()) == 0, lib.glam_last_error()
    ok = ok and torch.equal(res[1][5], want_xw) and torch.equal(res[1][6], want_a) and not torch.isnan(res[1][5]).any()
    # the embedding in front of the first application
    K = int(rng.choice([16, 32, 64]))
    xin, w0, b0 = r(N, K), r(C, K) * 0.4, r(C)
    img = torch.empty(lib.glam_ts_gemm_image_bytes(K, C) // 4, device=dev)
    assert lib.glam_ts_gemm_make_image(p(w0), K, 1, K, C, p(img), st()) == 0
    eact = 4 if train else int(rng.choice([0, 1]))
    s0 = torch.tensor([7 + case] + [0] * (ops.RNG_STATE_WORDS - 1), dtype=torch.int64, device=dev); e0 = torch.zeros(2, dtype=torch.int64, device=dev)
    s1, e1 = s0.clone(), e0.clone()
    o0, d0, o1, d1, xw1, a1 = f(N, C), f(N, C), f(N, C), f(N, C), f(N, HC), f(N, 8)
    if lib.glam_ts_gemm_rrelu_supported(K, C) == 1:
        if eact == 4:
            assert lib.glam_ts_gemm_rrelu(p(xin), K, K, p(img), p(b0), C, N, lo, hi, dp, p(s0), p(e0), p(o0), p(d0) if dp > 0 else None, st()) == 0
        elif eact == 1:
            assert lib.glam_ts_gemm_relu(p(xin), K, K, p(img), p(b0), p(o0), C, C, N, st()) == 0
        else:
            assert lib.glam_ts_gemm(p(xin), K, K, None, 0, 0, p(img), p(b0), p(o0), C, C, None, 0, 0, N, st()) == 0
        rows0 = d0 if (eact == 4 and dp > 0) else o0
        assert lib.glam_ts_gemm(p(rows0), C, C, None, 0, 0, p(nimg), None, p(want_xw), HC, HC, p(want_a), 8, 8, N, st()) == 0
        rc = lib.glam_ts_gemm_act_node(p(xin), K, K, p(img), p(b0), C, N, eact, lo, hi, dp, p(s1) if eact == 4 else None, p(e1) if eact == 4 else None,
                                       p(o1), p(d1) if (eact == 4 and dp > 0) else None, p(nfrag), HC, p(xw1), p(a1), st())
        assert rc == 0, lib.glam_last_error()
        ok = ok and torch.equal(o0, o1) and torch.equal(xw1, want_xw) and torch.equal(a1, want_a) and (eact != 4 or dp == 0 or torch.equal(d0, d1))
    bad += not ok
    print(("ok  " if ok else "BAD ") + f"case {case}: C={C} H={H} N={N} celu={celu} ident={ident} train={train} p={dp} act={act} K={K}", flush=True)
print(f"{n_cases - bad}/{n_cases} cases passed")
sys.exit(1 if bad else 0)
