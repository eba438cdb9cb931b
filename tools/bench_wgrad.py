"""Micro-benchmark of the batched weight-gradient kernels (run under rocprofv3 --kernel-trace, read tools/kstats.py):
bench_wgrad.py [pairs] [frames] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import kernels as K

P = int(sys.argv[1]) if len(sys.argv) > 1 else 7
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
H = W = 64
torch.manual_seed(0)
for (Ci, Co, ks) in [(144, 144, 3), (288, 144, 3), (144, 144, 1), (144, 576, 1)]:
    xs = [torch.randn(N, H, W, Ci, device="cuda").to(torch.bfloat16) for _ in range(P)]
    dys = [torch.randn(N, H, W, Co, device="cuda").to(torch.bfloat16) for _ in range(P)]
    dw = torch.zeros(Co, Ci, ks, ks, device="cuda")
    db = torch.zeros(Co, device="cuda")
    for _ in range(reps):
        K.conv_wgrad_batched(xs, dys, dw if ks == 3 else dw.reshape(Co, Ci), db, ks, N, H, W)
    torch.cuda.synchronize()
print("done")
