"""conv3x3 C -> C at the full config's recurrence size (M = 2 * 64 * 64 pixels: 64 tiles of the weight-streaming kernel on 256 CUs): the
weight-streaming kernel against the K-split kernel with 2 or 3 output-channel blocks per pixel tile.  Stream events around back-to-back launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vmg_amd import hip, kernels as K  # noqa: E402

reps = 200
for C in (112, 144):
    for N in (2, 4):
        H = W = 64
        x = torch.randn(N, H, W, C, device="cuda").to(torch.bfloat16)
        r = torch.randn(N, H, W, C, device="cuda").to(torch.bfloat16)
        w = torch.randn(C, C, 3, 3, device="cuda") * (C * 9) ** -0.5
        b = torch.randn(C, device="cuda")
        out = torch.empty(N, H, W, C, device="cuda", dtype=torch.bfloat16)
        variants = [("ws", K.pack_conv_weight_ws(w, cout_tiles=C // 16), 3)]
        for t in (3, 4, 5):
            variants.append((f"ksplit{t}", K.pack_conv_weight(w, torch.bfloat16, cout_tiles=t), 2))
        line = []
        for name, p, deep in variants:
            def run():
                K.conv_forward([x], p, b, N, H, W, alpha=0.1, res=r, out=out, deep=deep)
            for _ in range(10):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            line.append("%s %.1f us" % (name, e0.elapsed_time(e1) / reps * 1e3))
        print("C = %d, M = %d: " % (C, N * H * W) + " | ".join(line))
