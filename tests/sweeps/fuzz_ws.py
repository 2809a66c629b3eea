"""Randomised sweep of the warp-specialised TripletMessage kernels (csrc/triplet_ws*.hip) on molecular batches: against the oracle with
the fp64-twin bound, and against the general kernels (bit equality of the output and of d_x / d_weight_node / d_weight_scale / d_bias;
rounding-level agreement of d_weight_edge / d_weight_triplet_att).  Batch sizes from one molecule to thousands, every width of the
fused table (C = 33 .. 64), 1-4 heads, single-atom molecules mixed in.  usage: python tests/sweeps/fuzz_ws.py [n_cases] [seed]"""
import os, sys
os.environ.setdefault("GLAM_TORCH_EXT", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from glam_amd import _lib, layer, ops
from glam_amd.data import Batch, Data, synth_batch
import oracle.glam_oracle as O
from tests.conftest import assert_fp32_parity

dev = torch.device("cuda")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
names = ["x", "weight_node", "weight_edge", "weight_triplet_att", "weight_scale", "bias"]
bad = 0
for case in range(n_cases):
    B = int(rng.choice([1, 2, 3, 7, 16, 50, 200, 777, 1024, 3000]))
    H = int(rng.choice([1, 2, 3, 3, 3, 4]))
    cmax = 64 if H <= 3 else 48
    C = int(rng.integers(33, cmax + 1))
    cfg = dict(B=B, H=H, C=C, singles=bool(rng.random() < 0.3))
    try:
        b = synth_batch(B, seed=1000 + case)
        if cfg["singles"]:      # a few single-atom molecules (E = 0 for them: output row = bias)
            parts = [Data(torch.zeros(1, 1), torch.zeros(2, 0, dtype=torch.long), torch.zeros(0, 4)) for _ in range(int(rng.integers(1, 4)))]
            extra = Batch.from_data_list(parts)
            N0 = b.x.size(0)
            b = Data(torch.zeros(N0 + extra.x.size(0), 1), b.edge_index, b.edge_attr, batch=None)
        N = b.x.size(0)
        torch.manual_seed(case)
        conv = layer.TripletMessage(C, 4, heads=H)
        with torch.no_grad():
            conv.bias.normal_(0, 0.1)
        x0, cot = torch.randn(N, C), torch.randn(N, C)
        ps0 = [p.detach().clone() for p in conv.parameters()]
        ref = {}
        for dt in (torch.float32, torch.float64):
            xo = x0.to(dt).requires_grad_(True)
            ps = [p.to(dt).clone().requires_grad_(True) for p in ps0]
            o = O.triplet_message(xo, b.edge_index, b.edge_attr.to(dt), *ps, heads=H)
            ref[dt] = (o.detach(), torch.autograd.grad((o * cot.to(dt)).sum(), [xo] + ps))
        conv = conv.to(dev)
        ei, ea = b.edge_index.to(dev), b.edge_attr.to(dev)
        res = {}
        for mode in ("0", "auto"):
            ops.WS_ROUTE = mode
            x = x0.to(dev).requires_grad_(True)
            with _lib.kernel_timer(capacity=64) as kt:
                out = conv(x, ei, ea)
                gs = torch.autograd.grad(out, [x] + list(conv.parameters()), grad_outputs=cot.to(dev))
            res[mode] = (out, gs, [n for n, _, _ in kt.records()])
        ws_names = res["auto"][2]
        Cp = (C + 3) // 4 * 4
        fused = H * Cp + 8 <= 192                  # beyond: the wide-layer node (library GEMMs + general aggregate kernels)
        assert any("k_triplet_fwd_ws" in n for n in ws_names) == fused and any("k_triplet_bwd_src_ws" in n for n in ws_names) == fused, ws_names
        assert any("k_triplet_bwd_dst_ws" in n for n in ws_names) == (fused and H <= 3), ws_names
        b1ws = fused and H <= 3
        x3 = os.environ.get("GLAM_X3", "1") != "0"      # 3 x bf16 products in the warp-specialised kernels: rounding-level agreement
        if x3:
            err = (res["0"][0] - res["auto"][0]).abs().max().item()
            assert err <= 4e-6 * max(1.0, res["0"][0].abs().max().item()), f"out vs general {err:.2e}"
        else:
            assert torch.equal(res["0"][0], res["auto"][0]), "out differs from the general kernels"
        for n, a, c in zip(names, res["0"][1], res["auto"][1]):
            if x3 or (n in ("weight_edge", "weight_triplet_att") and b1ws):
                err = (a - c).abs().max().item()
                assert err <= 4e-6 * max(1.0, a.abs().max().item()), f"d_{n} vs general {err:.2e}"
            else:
                assert torch.equal(a, c), f"d_{n} differs from the general kernels ({(a - c).abs().max().item():.2e})"
        assert_fp32_parity(res["auto"][0], ref[torch.float64][0], ref[torch.float32][0], "out", out_tol=1e-5)
        for n, a, r64, r32 in zip(names, res["auto"][1], ref[torch.float64][1], ref[torch.float32][1]):
            assert_fp32_parity(a, r64, r32, "d_" + n)
        print("ok  ", cfg, f"N={N}", flush=True)
    except Exception as e:   # noqa: BLE001
        bad += 1
        print("FAIL", cfg, "->", type(e).__name__, str(e)[:300], flush=True)
print(f"{n_cases - bad}/{n_cases} cases passed")
sys.exit(1 if bad else 0)
