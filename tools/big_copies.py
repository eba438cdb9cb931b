"""Which ops issue the full-size strided copies / adds / cats of a train step?  One profiled step (CPU activity, shapes); prints every aten::copy_ /
aten::add / aten::add_ / aten::cat / aten::mul whose largest input has >= N elements (default 4 M; 0 lists every one, with aten::fill_ / zero_) with its chain of enclosing ops.
python tools/big_copies.py [N]"""
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
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    step(lrs, hrs)
    torch.cuda.synchronize()
THRESH = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
NAMES = ("aten::copy_", "aten::add", "aten::add_", "aten::cat", "aten::mul", "aten::mul_") + (("aten::fill_", "aten::zero_") if THRESH == 0 else ())
cnt = collections.Counter()
for ev in prof.events():
    if ev.name not in NAMES:
        continue
    shapes = [s for s in (ev.input_shapes or []) if s]
    big = 0
    for s in shapes:
        if isinstance(s, (list, tuple)) and s and all(isinstance(v, int) for v in s):
            n = 1
            for v in s:
                n *= v
            big = max(big, n)
        elif isinstance(s, (list, tuple)):  # (cat: a list of shapes)
            for t in s:
                if isinstance(t, (list, tuple)) and t and all(isinstance(v, int) for v in t):
                    n = 1
                    for v in t:
                        n *= v
                    big = max(big, n)
    if big < THRESH:
        continue
    chain, p = [], ev.cpu_parent
    while p is not None and len(chain) < 4:
        chain.append(p.name.replace("autograd::engine::evaluate_function: ", "")[:40])
        p = p.cpu_parent
    cnt[(ev.name, " <- ".join(chain) or "-", str(shapes[0] if shapes else "")[:40])] += 1
for (name, chain, shp), c in cnt.most_common(40 if THRESH else 90):
    print("%4d  %-12s %-40s %s" % (c, name, shp, chain))
