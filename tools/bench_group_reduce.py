"""Pooled sums (vmg_group_reduce: the GAP of the channel attention, 28 frames x 4096 pixels x 144 channels, bf16) with stream events; with the
diagnostics build VMG_GR_BLOCKS=n sets the grid.  python tools/bench_group_reduce.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import kernels as K

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
torch.manual_seed(0)
r = torch.randn(28, 64, 64, 144, device="cuda").to(torch.bfloat16)
for _ in range(3):
    K.group_reduce(r, 28, scale=1.0 / 4096)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    K.group_reduce(r, 28, scale=1.0 / 4096)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / reps * 1e3
print("group_reduce (zero-fill + kernel): %.1f us  (%.0f GB/s over the 33 MB input)" % (t, r.numel() * 2 / t / 1e3))
