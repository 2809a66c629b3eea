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
