#!/usr/bin/env python3
"""Secondary benchmark: full Architecture (mol_lin0 -> 3 x MessageBlock(_TripletMessage, GRU) -> GlobalPool5 ->
mol_flat -> lin_out1) fwd + bwd + Adam step on an ESOL-shaped batch (the reference's train iteration,
src_1gp/trainer.py:286-298).  Not the headline metric (bench.py is)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glam_amd import model, ops
from glam_amd.data import synth_batch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--norm", default="_None")
ap.add_argument("--block", default="_TripletMessage")
ap.add_argument("--readout", default="GlobalPool5")
ap.add_argument("--alpha", type=int, default=4, help="hid_dim_alpha (hidden width = 15 * alpha)")
ap.add_argument("--out-dim", type=int, default=1)
ap.add_argument("--loss", default="mse", choices=["mse", "bcel"], help="bcel: masked BCEWithLogits over labels >= 0 (trainer.py:244-245)")
ap.add_argument("--preset", default="relu", choices=["relu", "model_default", "run_default"],
                help="relu: deterministic ReLU / no dropout (parity configuration); model_default: Architecture() keyword defaults "
                     "(model.py:24-33: RReLU x 3, graph_do = end_do = Dropout(0.2), no graph norm) in train(); run_default: run.py:21-38 "
                     "(_NNConv unless --block is given, _PairNorm, flat_do = end_do = Dropout(0.2), RReLU x 3) in train()")
ap.add_argument("--no-graph", action="store_true")
ap.add_argument("--profile", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda")
torch.manual_seed(0)
if args.preset == "relu":
    net = model.Architecture(hid_dim_alpha=args.alpha, out_dim=args.out_dim, mol_block=args.block, message_steps=3, mol_readout=args.readout, graph_norm=args.norm,
                             graph_do="_None()", end_do="_None()", pre_act="ReLU", graph_act="ReLU", flat_act="ReLU").to(dev)
elif args.preset == "model_default":
    net = model.Architecture(hid_dim_alpha=args.alpha, out_dim=args.out_dim, mol_block=args.block, mol_readout=args.readout).to(dev).train()
else:
    blk = args.block if "--block" in sys.argv else "_NNConv"
    net = model.Architecture(hid_dim_alpha=args.alpha, out_dim=args.out_dim, mol_block=blk, mol_readout=args.readout, graph_norm="_PairNorm",
                             graph_do="_None()", flat_do="Dropout(0.2)", end_do="Dropout(0.2)", graph_res=1).to(dev).train()
    args.block, args.norm = blk, "_PairNorm"
b = synth_batch(args.batch, seed=0).to(dev)
y = b.y.view(-1) if args.out_dim == 1 else torch.randn(args.batch, args.out_dim, device=dev).view(-1)
# GLAM_ADAM=glam (default): glam_amd.optim.Adam, one HIP launch per step; torch: the library's fused multi-tensor optimizer
if os.environ.get("GLAM_ADAM", "glam") == "glam":
    from glam_amd import optim
    opt = optim.Adam(net.parameters(), lr=1e-3)
else:
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True, fused=os.environ.get("GLAM_FUSED_ADAM", "1") == "1")

if args.loss == "bcel":      # multi-task labels in {-1 (missing), 0, 1} (dataset.py:138)
    y = torch.randint(-1, 2, (args.batch * args.out_dim,), device=dev).float()

USE_GLAM_LOSS = os.environ.get("GLAM_LOSS", "glam") == "glam"     # glam_amd.loss: value + gradient in one launch; torch: the library's ops
if USE_GLAM_LOSS:
    from glam_amd import loss as glam_loss

def loss_of(out):
    if USE_GLAM_LOSS:
        return glam_loss.mse_loss(out, y) if args.loss == "mse" else glam_loss.bce_with_logits(out, y, masked=True)
    if args.loss == "mse":
        return torch.nn.functional.mse_loss(out, y)
    # mean over the valid labels only, without the data-dependent boolean indexing of the reference (not capturable)
    valid = (y >= 0).float()
    return (torch.nn.functional.binary_cross_entropy_with_logits(out, y.clamp(min=0), reduction="none") * valid).sum() / valid.sum().clamp(min=1)

ONE = torch.ones((), device=dev)      # the root gradient, kept across steps (loss.backward() alone launches a fill per step)

def body():
    opt.zero_grad(set_to_none=True)
    loss = loss_of(net(b).view(-1))
    loss.backward(gradient=ONE)
    opt.step()

side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): body()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
if args.profile:      # which torch ops (pads, copies, fills, adds, library GEMMs) a step still launches, by glam_amd call site
    import collections, traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    SKIP = ("empty", "view", "as_strided", "slice", "select", "detach", "alias", "t.", "transpose", "expand", "reshape", "unsqueeze",
            "squeeze", "narrow", "_unsafe_view", "permute", "_local_scalar_dense", "lift_fresh", "unbind", "split", "is_same_size")
    rows = collections.Counter()
    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if not any(name.startswith("aten." + k) for k in SKIP):
                fr = [f for f in traceback.extract_stack() if "glam_amd/" in f.filename or "tools/bench_model" in f.filename]
                site = " < ".join(f"{f.filename.split('/')[-1]}:{f.lineno}" for f in reversed(fr[-3:])) if fr else "(autograd engine)"
                shp = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), ())
                rows[(name, site, shp)] += 1
            return func(*args, **(kwargs or {}))
    gp = torch.cuda.CUDAGraph()          # inside a capture: the routes (and the glue) of the replayed step, not of an eager one
    with torch.cuda.graph(gp), Log():
        body()
    torch.cuda.synchronize()
    for (name, site, shp), c in sorted(rows.items(), key=lambda kv: (kv[0][0], -kv[1])):
        print(f"{c:4d}  {name:34s} {str(shp):18s} {site}")
    sys.exit(0)
g = None
if not args.no_graph:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): body()
step = g.replay if g is not None else body
for _ in range(10): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(args.steps): step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"workload": f"Architecture({args.block}, hid_dim_alpha={args.alpha}, 3 steps, {args.readout}, e_dim=1024, norm={args.norm}, "
                              f"out_dim={args.out_dim}, loss={args.loss}, preset={args.preset}) fwd+bwd+Adam, B={args.batch}",
                  "launch": "eager" if g is None else "hipGraph", "ms_per_step": dt / args.steps * 1e3,
                  "molecules_per_s": args.batch * args.steps / dt}))
