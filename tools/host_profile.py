"""Developer aid: where the HOST time of an eagerly issued full-model training step goes (cProfile over 300 steps at B = 32)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glam_amd import model, optim
from glam_amd.data import synth_batch

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = model.Architecture(mol_block="_TripletMessage").to(dev).train()
b = synth_batch(int(os.environ.get("B", "32")), seed=0).to(dev)
y = b.y.view(-1)
opt = optim.Adam(net.parameters(), lr=1e-3)

def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.mse_loss(net(b).view(-1), y)
    loss.backward()
    opt.step()

for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    step()
torch.cuda.synchronize()
print(f"eager step: {(time.perf_counter() - t0) / 300 * 1e3:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(35)
