"""Experiment: would one group per (node, head) beat one group per node?  Times the H=1 aggregate
kernels on a batch with 3x the nodes (same row bytes as H=3 on 1x), next to the real H=3 case."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from glam_amd import layer
from glam_amd.data import synth_batch

dev = torch.device("cuda:0")
for B, H in [(1024, 3), (3072, 1), (16384, 3), (49152, 1)]:
    torch.manual_seed(0)
    b = synth_batch(B, seed=0).to(dev)
    conv = layer.TripletMessage(60, 4, heads=H).to(dev)
    x = torch.randn(b.x.size(0), 60, device=dev)
    print(B, H, x.size(0), bench.time_kernels(conv, b, x, reps=100), flush=True)
