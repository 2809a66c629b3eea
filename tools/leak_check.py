import sys, os, gc
sys.path.insert(0, os.getcwd())
import torch
from glam_amd import model, ops
from glam_amd.data import synth_batch, synth_protein_batch
dev = torch.device("cuda")
torch.manual_seed(0)
two_tower = len(sys.argv) > 1 and sys.argv[1] == "dti"
if two_tower:
    net = model.ArchitectureDTI(hid_dim_alpha=3, graph_norm="_LayerNorm", graph_do="_None()", end_do="_None()", pre_act="ReLU", graph_act="ReLU",
                                flat_act="ReLU", end_act="ReLU").to(dev)
    cpu_pro = [synth_protein_batch(4, seed=s, n_min=30, n_max=120) for s in range(50)]
else:
    net = model.Architecture(hid_dim_alpha=3, mol_block="_NNConv", graph_norm="_PairNorm", mol_readout="Set2Set", graph_do="_None()",
                             end_do="_None()", pre_act="ReLU", graph_act="ReLU", flat_act="ReLU").to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
cpu_batches = [synth_batch(4 if two_tower else 16, seed=s) for s in range(50)]
# argv "routed": the model's graphed-callable route on (glam_amd.graphs.GraphedCallable keeps ONE private static copy per batch content, so
# the staging caches hold one entry per distinct content — 50 here — and stop growing); default: the eager path, whose caches must stay
# at the handful of live batches
routed = "routed" in sys.argv
net.graphed_call = routed
cap = 50 + 8 if routed else 8
marks = []
for step in range(1500):
    b = cpu_batches[step % 50].to(dev)            # a NEW device batch object every step (a shuffling loader)
    opt.zero_grad(set_to_none=True)
    out = net(b, cpu_pro[step % 50].to(dev)) if two_tower else net(b)
    torch.nn.functional.mse_loss(out.view(-1), b.y.view(-1)).backward()
    opt.step()
    if step % 500 == 499:
        torch.cuda.synchronize(); gc.collect()
        marks.append((torch.cuda.memory_allocated() >> 10, len(ops._GI_CACHE), len(ops._SP_CACHE), len(ops._PADDED), len(ops._ONEHOT_CACHE)))
        print("step", step + 1, "allocated KiB / graph-index / segment-ptr / padded / one-hot cache sizes:", marks[-1], flush=True)
assert marks[-1][0] <= marks[0][0] * 1.05 + 1024, "device memory grows"
assert all(m[1] <= cap and m[2] <= cap and m[3] <= 64 + cap and m[4] <= cap for m in marks), "host caches grow"
assert marks[-1][1:] == marks[0][1:] or not routed, "host caches still growing after the first 500 steps"
print("no growth")
