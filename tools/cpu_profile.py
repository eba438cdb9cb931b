"""cProfile of the host side of a few train steps (top cumulative / total-time entries)."""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vmg_amd.data import synthetic_clip, synthetic_target
from vmg_amd.train import TrainStep

dev = torch.device("cuda", 0)

model = bench.build_model(dev)
ts = TrainStep(model)
x = synthetic_clip(4, 7, 64, 64, seed=1, device=dev)
y = synthetic_target(x)
for _ in range(3):
    ts(x, y)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    ts(x, y)
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print("\n".join(l[:170] for l in s.getvalue().splitlines()[:45]))
