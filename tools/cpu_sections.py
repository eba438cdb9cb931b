"""Host-side time of the train step's sections (no device synchronisation inside a step): shows whether the step is
launch-bound and where the Python / autograd time goes.   python tools/cpu_sections.py [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vmg_amd.data import synthetic_clip, synthetic_target
from vmg_amd.train import TrainStep, charbonnier_edge_loss_hip, FlatAdamW
from vmg_amd import functional as FH

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
torch.backends.cudnn.benchmark = True
model = bench.build_model(dev)
ts = TrainStep(model)
x = synthetic_clip(4, 7, 64, 64, seed=1, device=dev)
y = synthetic_target(x)
for _ in range(3):
    ts(x, y)
torch.cuda.synchronize()
acc = {}
t_all0 = time.perf_counter()
for _ in range(steps):
    t = [time.perf_counter()]
    out = ts.model(x); t.append(time.perf_counter())
    loss = charbonnier_edge_loss_hip(out.float(), y.float(), ts.loss_args["eps"], ts.loss_args["aux_ratio"]); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    ts._flush(); t.append(time.perf_counter())
    if isinstance(ts.opt, FlatAdamW):
        ts.opt.advance(); ts.opt.launch(); FH.bump_weight_epoch()
    else:
        ts.opt.step()
    t.append(time.perf_counter())
    ts.opt.zero_grad(set_to_none=True); t.append(time.perf_counter())
    for name, a, b in zip(("forward", "loss", "backward", "flush", "optimizer", "zero_grad"), t[:-1], t[1:]):
        acc[name] = acc.get(name, 0.0) + (b - a)
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all0) / steps
print(f"wall {wall*1e3:.2f} ms/step; host time per section (ms): " + ", ".join(f"{k} {v/steps*1e3:.2f}" for k, v in acc.items()) + f"; host total {sum(acc.values())/steps*1e3:.2f}")
