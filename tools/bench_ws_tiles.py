"""conv3x3 144 -> 144 bf16 on the weight-streaming kernel: 144-channel blocks (one 159 KB workgroup per CU) vs 48-channel blocks (round 4: three
workgroups per tile, 80 KB each, two per CU) vs the K-split kernel, at the step's launch sizes.  Eight buffer sets are rotated so that operands
do not stay in the Infinity Cache; stream events around the batch.   python tools/bench_ws_tiles.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import hip, kernels as K

C = 144
dt = torch.bfloat16
torch.manual_seed(0)
w = torch.randn(C, C, 3, 3, device="cuda") * (C * 9) ** -0.5
b = torch.zeros(C, device="cuda")
packs = {"ws 9 tiles": (K.pack_conv_weight_ws(w, cout_tiles=9), 3), "ws 3 tiles x 3 blocks": (K.pack_conv_weight_ws(w, cout_tiles=3), 3),
         "k-split": (K.pack_conv_weight(w, dt, cout_tiles=3), 2)}
for (N, H, W) in [(8, 64, 64), (28, 64, 64), (2, 256, 448), (8, 128, 128)]:
    M = N * H * W
    nset = 8 if M <= 131072 else 4
    xs = [torch.randn(N, H, W, C, device="cuda").to(dt) for _ in range(nset)]
    rs = [torch.randn(N, H, W, C, device="cuda").to(dt) for _ in range(nset)]
    outs = [torch.empty(N, H, W, C, device="cuda", dtype=dt) for _ in range(nset)]
    line = [f"M = {M:7d} ({N}x{H}x{W}):"]
    for name, (pw, deep) in packs.items():
        def run(i):
            K.conv_forward([xs[i % nset]], pw, b, N, H, W, alpha=0.1, res=rs[i % nset], out=outs[i % nset], deep=deep)
        for i in range(8):
            run(i)
        torch.cuda.synchronize()
        reps = 64
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        line.append(f"{name} {us:7.1f} us ({2.0 * M * C * C * 9 / us / 1e6:6.1f} TFLOP/s)")
    print("  ".join(line))
