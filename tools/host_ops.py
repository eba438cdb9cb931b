"""Host (CPU) time of a train step by operator, forward and backward (torch.profiler, CPU activity only): which autograd nodes / wrappers
keep the host busy.   python tools/host_ops.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

dev = torch.device("cuda", 0)
from vmg_amd.data import synthetic_clip, synthetic_target
from vmg_amd.train import TrainStep
model = bench.build_model(dev)
step = TrainStep(model, lr=2e-4, betas=(0.9, 0.99), aux=True, aux_ratio=0.005, distributed=False)
lrs = synthetic_clip(4, 7, 64, 64, seed=1234, device=dev)
hrs = synthetic_target(lrs, seed=4321)
for _ in range(3):
    step(lrs, hrs)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(2):
        step(lrs, hrs)
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages():
    rows.append((ev.self_cpu_time_total / 2e3, ev.cpu_time_total / 2e3, ev.count / 2, ev.key))
rows.sort(reverse=True)
print("%9s %9s %7s  %s" % ("self ms", "total ms", "calls", "op  (per step)"))
for r in rows[:45]:
    print("%9.2f %9.2f %7.0f  %s" % r)
