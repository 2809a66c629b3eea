"""Two-tower (ligand + protein) step, BASELINE config 5 shape: B pairs, proteins of 200..800 residues; fwd+bwd+Adam, hipGraph."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glam_amd import model, optim
from glam_amd.data import synth_batch, synth_protein_batch

_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(_pos[0]) if len(_pos) > 0 else 32
NORM = _pos[1] if len(_pos) > 1 else "_None"
dev = torch.device("cuda")
torch.manual_seed(0)
net = model.ArchitectureDTI(graph_norm=NORM, graph_do="_None()", end_do="_None()", pre_act="ReLU", graph_act="ReLU", flat_act="ReLU", end_act="ReLU").to(dev)
mol, pro = synth_batch(B, seed=0).to(dev), synth_protein_batch(B, seed=1, n_min=200, n_max=800).to(dev)
y = torch.randn(B, device=dev)
opt = (optim.Adam(net.parameters(), lr=1e-3) if os.environ.get("GLAM_ADAM", "glam") == "glam"
           else torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True, fused=True))

# GLAM_LOSS=glam (default): glam_amd.loss (what get_loss('mse') of the reference's trainer maps to: value + gradient in one launch); torch: F.mse_loss
from glam_amd import loss as glam_loss
loss_fn = glam_loss.mse_loss if os.environ.get("GLAM_LOSS", "glam") == "glam" else torch.nn.functional.mse_loss
ONE = torch.ones((), device=dev)      # the root gradient, kept across steps (loss.backward() alone launches a fill per step)

def body():
    opt.zero_grad(set_to_none=True)
    loss_fn(net(mol, pro).view(-1), y).backward(gradient=ONE)
    opt.step()

side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): body()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
if "--profile" in sys.argv:      # which torch ops a captured step still launches, by glam_amd call site (see bench_model.py --profile)
    import collections, traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    SKIP = ("empty", "view", "as_strided", "slice", "select", "detach", "alias", "t.", "transpose", "expand", "reshape", "unsqueeze",
            "squeeze", "narrow", "_unsafe_view", "permute", "_local_scalar_dense", "lift_fresh", "unbind", "split", "is_same_size")
    rows = collections.Counter()
    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if not any(name.startswith("aten." + k) for k in SKIP):
                fr = [f for f in traceback.extract_stack() if "glam_amd/" in f.filename or "tools/bench_dti" in f.filename]
                site = " < ".join(f"{f.filename.split('/')[-1]}:{f.lineno}" for f in reversed(fr[-3:])) if fr else "(autograd engine)"
                shp = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), ())
                rows[(name, site, shp)] += 1
            return func(*args, **(kwargs or {}))
    gp = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gp), Log():
        body()
    torch.cuda.synchronize()
    for (name, site, shp), c in sorted(rows.items(), key=lambda kv: (kv[0][0], -kv[1])):
        print(f"{c:4d}  {name:34s} {str(shp):18s} {site}")
    sys.exit(0)
for mode in ("eager", "hipGraph"):
    step = body
    if mode == "hipGraph":
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): body()
        step = g.replay
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(json.dumps({"workload": f"ArchitectureDTI defaults (_NNConv ligand, _GCNConv protein, norm={NORM}), B={B} pairs, protein nodes={pro.x.size(0)}",
                      "launch": mode, "ms_per_step": dt * 1e3, "pairs_per_s": B / dt}), flush=True)
