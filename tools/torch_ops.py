"""Which Python lines issue the torch (non-HIP-library) device ops of a train step?  One profiled step with stacks; prints
(op, count, innermost vmg_amd / bench frame).   python tools/torch_ops.py"""
import collections, os, sys
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
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    step(lrs, hrs)
    torch.cuda.synchronize()
cnt = collections.Counter()
LEAF = ("aten::add_", "aten::add", "aten::zeros", "aten::zero_", "aten::fill_", "aten::copy_", "aten::mul", "aten::mul_", "aten::cat", "aten::stack",
        "aten::clone", "aten::contiguous", "aten::flip", "aten::sum", "aten::to", "aten::_to_copy", "aten::zeros_like", "aten::empty_like", "aten::div",
        "aten::sub", "aten::index", "aten::where", "aten::addmm", "aten::mm")
for ev in prof.events():
    if ev.name in LEAF and (ev.cpu_parent is None or not ev.cpu_parent.name.startswith("aten::")):
        frame = "?"
        for f in ev.stack or []:
            if "vmg_amd/" in f or "bench.py" in f:
                frame = f.split("/root/repo/")[-1] if "/root/repo/" in f else f
                break
        if frame == "?":
            frame = "(autograd engine) parent=" + (ev.cpu_parent.name[:60] if ev.cpu_parent is not None else "-")
        cnt[(ev.name, frame[:110])] += 1
for (name, frame), c in cnt.most_common(70):
    print("%5d  %-18s %s" % (c, name, frame))
