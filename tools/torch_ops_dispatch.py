"""Which Python lines issue the torch (non-HIP-library) device ops of a train step?  A TorchDispatchMode sees every aten op of one step (forward,
backward, update) with the Python stack of the moment; prints (count, op, output elements, innermost vmg_amd frame or the autograd node).
   python tools/torch_ops_dispatch.py [train|train_full|train_swin]"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
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

NO_KERNEL = ("view", "reshape", "permute", "transpose", "select", "slice", "narrow", "unsqueeze", "squeeze", "empty", "as_strided", "detach", "alias", "expand",
             "unbind", "chunk", "split", "t.", "unflatten", "flatten", "_unsafe_view", "item", "_local_scalar_dense", "lift_fresh", "movedim", "set_", "sym_",
             "is_same_size", "stride", "size", "numel", "dim", "storage_offset", "record_stream", "is_pinned", "_reshape_alias", "new_empty", "resize_")
cnt = collections.Counter()
elems = {}


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func).replace("aten.", "")
        if any(name.startswith(v) for v in NO_KERNEL):
            return out
        frame = None
        for f in reversed(traceback.extract_stack()[:-1]):
            if "/vmg_amd/" in f.filename or f.filename.endswith("bench.py"):
                frame = "%s:%d %s" % (f.filename.split("/vmg_amd/")[-1] if "/vmg_amd/" in f.filename else "bench.py", f.lineno, f.name)
                break
        key = (name, frame or "(autograd engine: no Python frame)")
        cnt[key] += 1
        t = out[0] if isinstance(out, (tuple, list)) and out else out
        if isinstance(t, torch.Tensor):
            elems[key] = max(elems.get(key, 0), t.numel())
        return out


with Spy():
    step(lrs, hrs)
torch.cuda.synchronize()
print("count  op                                 max elements  where")
for (name, frame), c in cnt.most_common(90):
    print("%5d  %-34s %12d  %s" % (c, name, elems.get((name, frame), 0), frame))
print("total ops with a kernel:", sum(cnt.values()))
