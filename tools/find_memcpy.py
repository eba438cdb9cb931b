"""Which torch ops of a train step issue device-to-device memcpys (graph memcpy nodes)?"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from vmg_amd.data import synthetic_clip, synthetic_target
from vmg_amd.train import TrainStep
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
step = TrainStep(model, lr=2e-4, betas=(0.9, 0.99), aux=True, aux_ratio=0.005, distributed=False)
lrs = synthetic_clip(4, 7, 64, 64, seed=1234, device=dev)
hrs = synthetic_target(lrs, seed=4321)
for _ in range(3):
    step(lrs, hrs)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(lrs, hrs)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    ks = getattr(ev, "kernels", None) or []
    if any("emcpy" in k.name or "copyBuffer" in k.name for k in ks):
        chain, p = [], ev
        while p is not None and len(chain) < 5:
            chain.append(p.name)
            p = p.cpu_parent
        cnt[" <- ".join(chain)] += 1
for k, c in cnt.most_common(30):
    print(c, k[:200])
