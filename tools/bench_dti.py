"""Two-tower (ligand + protein) step, BASELINE config 5 shape: B pairs, proteins of 200..800 residues; fwd+bwd+Adam, hipGraph."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glam_amd import model, optim
from glam_amd.data import synth_batch, synth_protein_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NORM = sys.argv[2] if len(sys.argv) > 2 else "_None"
dev = torch.device("cuda")
torch.manual_seed(0)
net = model.ArchitectureDTI(graph_norm=NORM, graph_do="_None()", end_do="_None()", pre_act="ReLU", graph_act="ReLU", flat_act="ReLU", end_act="ReLU").to(dev)
mol, pro = synth_batch(B, seed=0).to(dev), synth_protein_batch(B, seed=1, n_min=200, n_max=800).to(dev)
y = torch.randn(B, device=dev)
opt = (optim.Adam(net.parameters(), lr=1e-3) if os.environ.get("GLAM_ADAM", "glam") == "glam"
           else torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True, fused=True))

def body():
    opt.zero_grad(set_to_none=True)
    torch.nn.functional.mse_loss(net(mol, pro).view(-1), y).backward()
    opt.step()

side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): body()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
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
