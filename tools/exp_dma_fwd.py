"""Experiment: the software-pipelined forward aggregate (glam_triplet_fwd_ell: k_triplet_fwd_pipe, or k_triplet_fwd_dma with
GLAM_ELL_VARIANT=dma) vs k_triplet_fwd: bit equality + per-dispatch durations.  usage: exp_dma_fwd.py [B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glam_amd import _lib, layer, ops
from glam_amd.data import synth_batch

dev = torch.device("cuda:0")
lib = _lib.load()
p, st = _lib.ptr, _lib.stream
torch.manual_seed(0)
conv = layer.TripletMessage(60, 4).to(dev)
for B in [int(v) for v in (sys.argv[1:] or ["64", "1024", "16384"])]:
    b = synth_batch(B, seed=7 if B > 1024 else 0).to(dev)
    N, E = b.x.size(0), b.edge_index.size(1)
    x = torch.randn(N, 60, device=dev)
    gi = ops.graph_index(b.edge_index, N)
    with torch.no_grad():
        Wn, Wa, We, M, Ws, Cp, Dp = conv._staged_weights()
        xw, a_ij = (x @ Wn).contiguous(), (x @ Wa).contiguous()
    H = 3
    aggr0, stats0 = torch.empty(N, H * Cp, device=dev), torch.empty(N, 8, device=dev)
    aggr1, stats1 = torch.full((N, H * Cp), 7.0, device=dev), torch.full((N, 8), 7.0, device=dev)
    ell_s, ell_e = torch.empty(N, 4, dtype=torch.int32, device=dev), torch.empty(N, 4, dtype=torch.int32, device=dev)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.glam_ell_build(p(gi.rowptr), p(gi.src), p(gi.eid), N, p(ell_s), p(ell_e), p(ovf), st()), "ell")
    assert int(ovf.item()) == 0

    def ref():
        _lib.check(lib.glam_triplet_fwd(p(xw), p(a_ij), p(b.edge_attr), p(We), p(M), p(gi.rowptr), p(gi.src), p(gi.eid), N, E, H, Cp, Dp, 1, 0.2,
                                        p(aggr0), p(stats0), st()), "fwd")

    def dma(grid=0, onehot=0):
        _lib.check(lib.glam_triplet_fwd_ell(p(xw), p(a_ij), p(b.edge_attr), p(We), p(M), p(ell_s), p(ell_e), N, E, H, Cp, Dp, 0.2, onehot, p(aggr1),
                                            p(stats1), grid, st()), "fwd_ell")
    ref(); torch.cuda.synchronize()
    for oh in (0, 1):
        aggr1.fill_(7.0); stats1.fill_(7.0)
        dma(0, oh); torch.cuda.synchronize()
        print(f"B={B} N={N} onehot={oh}: aggr equal {torch.equal(aggr0, aggr1)} stats equal {torch.equal(stats0, stats1)} max|d| {(aggr0 - aggr1).abs().max().item():.2e}")
    for name, fn in [("k_triplet_fwd", ref)] + [(f"dma grid={g} oh={oh}", (lambda g=g, oh=oh: dma(g, oh))) for g in (512,) for oh in (0, 1)]:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        with _lib.kernel_timer(64) as kt:
            for _ in range(20):
                fn()
        torch.cuda.synchronize()
        us = [r[2] for r in kt.records()]
        nbytes = 4 * (2 * N * 180 + E * 4 + 2 * E + (N + 1) + 4 * N * 3)
        print(f"   {name:18s} avg {sum(us)/len(us):8.2f} us  min {min(us):8.2f}   {nbytes / (sum(us)/len(us)) / 1e3 / 8000:.3f} of 8 TB/s")
