"""3x3 weight gradients with few output channels at the bench batch's HR sizes (conv_last: 64 -> 3 on 28 x 256 x 256 pixels; dY a 3-channel
slice of an 8-channel buffer), bf16, timed with stream events: python tools/bench_small_wgrad.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import kernels as K

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.manual_seed(0)
for (N, H, W, Ci, Co) in [(28, 256, 256, 64, 3), (28, 128, 128, 64, 3), (4, 256, 256, 64, 3), (28, 256, 256, 64, 16)]:
    x = torch.randn(N, H, W, Ci, device="cuda").to(torch.bfloat16)
    dyb = torch.randn(N, H, W, (Co + 7) // 8 * 8, device="cuda").to(torch.bfloat16)
    dy = dyb[..., :Co]
    dw = torch.zeros(Co, Ci, 3, 3, device="cuda")
    db = torch.zeros(Co, device="cuda")
    for _ in range(2):
        K.conv_wgrad_batched([x], [dy], dw, db, 3, N, H, W)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        K.conv_wgrad_batched([x], [dy], dw, db, 3, N, H, W)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    mb = (x.numel() + dyb.numel()) * 2 / 1e6
    print("N %2d  %3dx%3d  %2d -> %2d : %8.1f us   (%.0f MB read once = %.0f GB/s)" % (N, H, W, Ci, Co, us, mb, mb / us * 1e3), flush=True)
