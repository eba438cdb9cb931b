"""HRconv (3x3, 64 -> 64, bias + LeakyReLU) on 28 x 256 x 256 pixels, bf16: the general kernel (deep 0) and the weights-stationary one (deep 6);
read the device durations with tools/prof_read.py on a rocprofv3 kernel trace (groups of 13 launches: python tools/prof_read.py <dir> conv_ 13)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import hip, kernels as K
N, H, W = 28, 256, 256
torch.manual_seed(0)
for ci, co in ((64, 64), (8, 64)):
    x = torch.randn(N, H, W, ci, device="cuda").to(torch.bfloat16)
    w = torch.randn(co, ci, 3, 3, device="cuda") / (ci * 9) ** 0.5
    b = torch.randn(co, device="cuda")
    out = torch.empty(N, H, W, co, device="cuda", dtype=torch.bfloat16)
    pw = K.pack_conv_weight(w, torch.bfloat16, cout_tiles=co // 16)
    for deep in (0, 6):
        for _ in range(13):
            K.conv_forward([x], pw, b, N, H, W, act=hip.ACT_LRELU, slope=0.1, out=out, deep=deep)
        torch.cuda.synchronize()
print("done")
