import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """One committed fixture: inputs, parameters, output, cotangent and gradients captured from the
    reference (see oracle/gen_goldens.py)."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
        self.name = name
        self.meta = json.loads(str(z["meta"]))
        self.inputs = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in.")}
        self.params = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")}
        self.grads = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("grad.")}
        self.out = torch.from_numpy(z["out"])
        self.cot = torch.from_numpy(z["cot"]) if "cot" in z.files else None


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLD) if f.startswith(prefix) and f.endswith(".npz"))


@pytest.fixture(scope="session")
def device():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


def assert_close(a, b, tol, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    if a.numel() == 0:
        return
    err = (a - b).abs().max().item()
    scale = max(1.0, b.abs().max().item())
    assert err <= tol * scale, f"{what}: max|d|={err:.3e} > {tol:.1e} * {scale:.3g}"


EPS32 = 2.0 ** -23


def assert_fp32_parity(got, ref64, ref32, what="", k=8.0, out_tol=None):
    """fp32 parity bound derived from the fp64 twin of the oracle, per quantity.

    ``noise = max|ref32 - ref64|`` is the rounding error the fp32 ORACLE itself makes on this very quantity (same inputs,
    same algorithm, fp64 vs fp32): it grows with the depth of the computation and with the length of the sums behind the
    quantity (a weight gradient at B = 1024 sums 20 k rows), which is exactly what a fixed tolerance ladder guessed at.  An
    fp32 implementation that contracts in another order (MFMA tiles, CSR segment sums, fixed-order block partials) draws a
    different sample of the same error, so it must sit within ``k * max(noise, 4 ulp(max(scale, 1)))`` of the fp64 value; k = 8 is
    the slack for max-norms of two independent samples over up to ~1e6 elements.  The floor is taken at unit scale at least:
    unit-scale intermediates (softmax weights in [0, 1], normalised activations) carry their own ulp-level errors into every
    downstream quantity whatever that quantity's magnitude (a d_att entry of scale 0.1 behind a softmax weight of scale 1), and a
    single fp32 sample of 100 elements underestimates that floor.  ``out_tol`` adds BASELINE.json's absolute bar
    for forward outputs (1e-5, scaled by max(1, |ref|)).  Returns (err, bound) for reporting."""
    g = got.detach().cpu().double()
    r64, r32 = ref64.detach().cpu().double(), ref32.detach().cpu().double()
    assert g.shape == r64.shape, f"{what}: shape {tuple(g.shape)} vs {tuple(r64.shape)}"
    if g.numel() == 0:
        return 0.0, 0.0
    scale = r64.abs().max().item()
    noise = (r32 - r64).abs().max().item()
    bound = k * max(noise, 4 * EPS32 * max(scale, 1.0))
    err = (g - r64).abs().max().item()
    if os.environ.get("GLAM_PARITY_REPORT"):      # developer aid: every check's error against its bound (pytest -s)
        print(f"[parity] {what}: err {err:.3e} bound {bound:.3e} ({err / bound:.2f})", flush=True)
    assert err <= bound, f"{what}: max|d| = {err:.3e} > {k:g} x fp64-twin noise floor = {bound:.3e} (noise {noise:.3e}, scale {scale:.3g})"
    if out_tol is not None:
        assert err <= out_tol * max(1.0, scale), f"{what}: max|d| = {err:.3e} > {out_tol:.0e} * max(1, {scale:.3g})"
    return err, bound


def assert_twin_parity(run, got_out, got_grads, what, names=None, out_tol=1e-5, k=8.0):
    """``run(dtype) -> (out, [grads])`` evaluates the ORACLE on the test's inputs in the given precision (CPU).  The HIP results must
    sit within the bound the oracle's own fp64 twin sets (``assert_fp32_parity``); the forward output additionally within
    BASELINE's 1e-5.  Replaces the fixed 2e-5 ... 2e-4 ladders: the bound now follows the depth / sum length of each quantity."""
    o32, g32 = run(torch.float32)
    o64, g64 = run(torch.float64)
    assert_fp32_parity(got_out, o64, o32, what + " out", k=k, out_tol=out_tol)
    names = names or [str(i) for i in range(len(g64))]
    for n, a, r64, r32 in zip(names, got_grads, g64, g32):
        if r64 is None:
            continue
        assert_fp32_parity(a, r64, r32, f"{what} grad.{n}", k=k)


def assert_golden_parity(run64, g, got_out, got_grads, what, names, out_key=None, out_tol=1e-5, k=8.0):
    """A golden fixture IS the reference's fp32 sample of a quantity; ``run64() -> (out, [grads])`` evaluates the oracle in fp64 on the
    fixture's inputs (its fp64 twin: the oracle is pinned to the reference by oracle/gen_goldens.py).  The HIP result must sit within
    the fp64-twin bound of ``assert_fp32_parity``; ``names`` index ``g.grads``; ``None`` entries of the twin are skipped."""
    o64, g64 = run64()
    ref_out = g.out if out_key is None else g.grads[out_key]
    if got_out is not None:
        assert_fp32_parity(got_out, o64, ref_out, what + " out", k=k, out_tol=out_tol)
    for n, a, r64 in zip(names, got_grads, g64):
        if r64 is None:
            continue
        assert_fp32_parity(a, r64, g.grads[n], f"{what} grad.{n}", k=k)
