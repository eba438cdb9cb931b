"""Ablation of the weights-stationary conv (diagnostics build, VMG_CONV_DBG bits: 1 no halo copies, 32 no K loop, 64 no epilogue, 8 no stores):
stream-event time of HRconv 64 -> 64 on 28 x 256 x 256 per bit set.  VMG_HIP_LIB=vmg_amd/libvmg_hip_diag.so python tools/wstat_ablate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import hip, kernels as K
N, H, W = 28, 256, 256
x = torch.randn(N, H, W, 64, device="cuda").to(torch.bfloat16)
w = torch.randn(64, 64, 3, 3, device="cuda") / 24
b = torch.randn(64, device="cuda")
out = torch.empty(N, H, W, 64, device="cuda", dtype=torch.bfloat16)
pw = K.pack_conv_weight(w, torch.bfloat16, cout_tiles=4)
for bits in (0, 8, 64, 32, 1, 33, 97, 96, 65):
    os.environ["VMG_CONV_DBG"] = str(bits)
    for _ in range(3):
        K.conv_forward([x], pw, b, N, H, W, act=hip.ACT_LRELU, slope=0.1, out=out, deep=6)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        K.conv_forward([x], pw, b, N, H, W, act=hip.ACT_LRELU, slope=0.1, out=out, deep=6)
    e1.record()
    torch.cuda.synchronize()
    print("dbg %3d: %7.1f us" % (bits, e0.elapsed_time(e1) * 100), flush=True)
