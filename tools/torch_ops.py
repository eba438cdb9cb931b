"""Which Python lines issue the torch (non-HIP-library) device ops of a train step?  One profiled step with stacks; prints
(op, count, innermost vmg_amd / bench frame).   python tools/torch_ops.py [train|train_full|train_swin]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

dev = torch.device("cuda", 0)
from vmg_amd.data import synthetic_clip, synthetic_target
from vmg_amd.train import TrainStep
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "train"]
model = bench.build_model(dev, wl)
step = TrainStep(model, lr=2e-4, betas=(0.9, 0.99), aux=True, aux_ratio=0.005, distributed=False)
lrs = synthetic_clip(wl["batch"], wl["frames"], wl["size"], wl["size"], seed=1234, device=dev)
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
VIEWS = ("aten::view", "aten::reshape", "aten::permute", "aten::transpose", "aten::select", "aten::slice", "aten::narrow", "aten::unsqueeze", "aten::squeeze",
         "aten::empty", "aten::empty_like", "aten::empty_strided", "aten::as_strided", "aten::detach", "aten::alias", "aten::expand", "aten::unbind", "aten::chunk",
         "aten::split", "aten::t", "aten::unflatten", "aten::flatten", "aten::_unsafe_view", "aten::view_as", "aten::item", "aten::_local_scalar_dense", "aten::is_nonzero",
         "aten::lift_fresh", "aten::movedim", "aten::split_with_sizes", "aten::result_type", "aten::new_empty", "aten::resolve_conj", "aten::resolve_neg", "aten::set_")
for ev in prof.events():
    # every top-level aten op (not called from another aten op) that is not a pure view / allocation
    if ev.name.startswith("aten::") and ev.name not in VIEWS and (ev.cpu_parent is None or not ev.cpu_parent.name.startswith("aten::")):
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
