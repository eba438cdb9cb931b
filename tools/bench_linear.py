"""1x1 / Linear kernels in isolation (read tools/kstats.py on a rocprofv3 kernel trace): bench_linear.py [M] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import hip, kernels as K
M = int(sys.argv[1]) if len(sys.argv) > 1 else 114688
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
torch.manual_seed(0)
x = torch.randn(M, 144, device="cuda").to(torch.bfloat16)
w = torch.randn(144, 144, device="cuda") / 12
b = torch.randn(144, device="cuda")
out = torch.empty(1, 1, M, 144, device="cuda", dtype=torch.bfloat16)
for tiles, deep in ((9, 0), (5, 4), (3, 4)):
    pw = K.pack_conv_weight(w, torch.bfloat16, cout_tiles=tiles)
    for _ in range(reps):
        K.conv_forward([x], pw, b, 1, 1, M, act=hip.ACT_RELU, out=out, deep=deep)
    torch.cuda.synchronize()
print("done")
