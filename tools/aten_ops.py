"""Developer aid: which torch (aten) ops a full-model training step still issues, with call counts per step and stack hints."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from glam_amd import model, ops, optim, loss as glam_loss
ops.USE_TORCH_EXT = False      # the route a captured step takes (Python nodes with the gradient carry)
from glam_amd.data import synth_batch

preset = sys.argv[1] if len(sys.argv) > 1 else "model_default"
dev = torch.device("cuda")
torch.manual_seed(0)
if preset == "relu":
    net = model.Architecture(mol_block="_TripletMessage", graph_do="_None()", end_do="_None()", pre_act="ReLU", graph_act="ReLU", flat_act="ReLU").to(dev)
elif preset == "run_default":
    net = model.Architecture(mol_block="_NNConv", graph_norm="_PairNorm", graph_do="_None()", flat_do="Dropout(0.2)", end_do="Dropout(0.2)").to(dev).train()
else:
    net = model.Architecture(mol_block="_TripletMessage").to(dev).train()
b = synth_batch(1024, seed=0).to(dev)
y = b.y.view(-1)
opt = optim.Adam(net.parameters(), lr=1e-3)

def body():
    opt.zero_grad(set_to_none=True)
    loss = glam_loss.mse_loss(net(b).view(-1), y)
    loss.backward()
    opt.step()

for _ in range(3):
    body()
torch.cuda.synchronize()
R = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(R):
        body()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=4) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
print(f"preset={preset}: aten ops with device time, per step")
for e in rows[:40]:
    stack = " <- ".join(s.split("/")[-1] for s in e.stack[:3]) if e.stack else ""
    print(f"{e.key:32s} n/step={e.count / R:5.1f} dev_us/step={e.device_time_total / R:8.1f}   {stack[:150]}")

# second view: the same ops grouped by input shapes (tells parameter-gradient accumulation from activation-gradient sums)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof2:
    for _ in range(R):
        body()
    torch.cuda.synchronize()
rows = [e for e in prof2.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
print("by input shape:")
for e in rows[:40]:
    print(f"{e.key:28s} n/step={e.count / R:5.1f} dev_us/step={e.device_time_total / R:8.1f}   {str(e.input_shapes)[:120]}")
