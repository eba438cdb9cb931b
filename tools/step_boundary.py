"""Host time between the end of one train step and the first launches of the next (the GPU is idle across the step boundary)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import bench
from vmg_amd.data import synthetic_clip, synthetic_target
from vmg_amd.train import TrainStep
import vmg_amd.model as M

dev = torch.device("cuda", 0)
model = bench.build_model(dev)
step = TrainStep(model, lr=2e-4, betas=(0.9, 0.99), aux=True, aux_ratio=0.005, distributed=False)
lrs = synthetic_clip(4, 7, 64, 64, seed=1234, device=dev)
hrs = synthetic_target(lrs, seed=4321)
for _ in range(3):
    step(lrs, hrs)
torch.cuda.synchronize()
marks = []
orig_interp, orig_flow, orig_fwd = F.interpolate, model.compute_flow, type(model)._forward
def interp(*a, **k):
    marks.append(("interpolate", time.perf_counter()))
    return orig_interp(*a, **k)
def flow(x):
    marks.append(("compute_flow begin", time.perf_counter()))
    r = orig_flow(x)
    marks.append(("compute_flow end", time.perf_counter()))
    return r
M.F.interpolate = interp
model.compute_flow = flow
for i in range(3):
    t0 = time.perf_counter()
    marks.append(("step begin", t0))
    out = model(lrs)
    marks.append(("forward end", time.perf_counter()))
    from vmg_amd.train import charbonnier_edge_loss_hip
    loss = charbonnier_edge_loss_hip(out.float(), hrs.float(), step.loss_args["eps"], step.loss_args["aux_ratio"])
    marks.append(("loss end", time.perf_counter()))
    loss.backward()
    marks.append(("backward end", time.perf_counter()))
    step._flush()
    step.opt.advance(); step.opt.launch()
    import vmg_amd.functional as FH
    FH.bump_weight_epoch(); FH.repack_all()
    marks.append(("optimizer+repack end", time.perf_counter()))
    step.opt.zero_grad(set_to_none=True)
    marks.append(("zero_grad end", time.perf_counter()))
torch.cuda.synchronize()
prev = None
for name, t in marks:
    print("%-24s +%8.1f us" % (name, (t - prev) * 1e6 if prev else 0.0))
    prev = t
