import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, collections
import bench
from vmg_amd import kernels as K
from vmg_amd.data import synthetic_clip, synthetic_target
from vmg_amd.train import TrainStep
dev = torch.device("cuda", 0)
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "train"]
model = bench.build_model(dev, wl)
ts = TrainStep(model)
x = synthetic_clip(wl["batch"], wl["frames"], wl["size"], wl["size"], seed=1, device=dev)
y = synthetic_target(x)
for _ in range(2):
    ts(x, y)
cnt = collections.Counter()
orig = K.act_backward
def wrap(dy, ref, act, slope, alpha):
    cnt[(tuple(dy.shape), act)] += 1
    return orig(dy, ref, act, slope, alpha)
K.act_backward = wrap
ts(x, y)
torch.cuda.synchronize()
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(v, k)
print("total act_backward calls:", sum(cnt.values()))
