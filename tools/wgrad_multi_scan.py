"""Per-problem cost of vmg_conv_wgrad3_multi as a function of the number of problems per launch (7 pairs x 8 frames of 64x64, 144 -> 144).
Run under rocprofv3 --kernel-trace --stats with the number of problems as argument."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import kernels as K

nprob = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = W = 64
C = 144
torch.manual_seed(0)
xs = [torch.randn(8, H, W, C, device="cuda").to(torch.bfloat16) for _ in range(7)]
dys = [torch.randn(8, H, W, C, device="cuda").to(torch.bfloat16) for _ in range(7)]
probs = [(xs, dys, torch.zeros(C, C, 3, 3, device="cuda"), torch.zeros(C, device="cuda"), 1.0) for _ in range(nprob)]
for _ in range(10):
    K.conv_wgrad3_multi(probs, 8, H, W)
torch.cuda.synchronize()
print("done", nprob)
