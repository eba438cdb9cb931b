"""Grouped 3x3 convolution (Mlp_cnn.fc1 of the full configuration: C -> 6C, 4 groups) forward + backward at its four stage sizes (one 7-frame clip
per GPU): G launches on the parameter's group slices (round 3) vs ONE launch on the dense block-diagonal pack (round 4, functional.GROUPED_DENSE).
Stream events around `reps` forward + backward pairs (weight gradients included).   python tools/bench_grouped_conv.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import hip, functional as FH

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dt = torch.bfloat16
for (N, H, W, C) in [(7, 64, 64, 112), (7, 32, 32, 224), (7, 16, 16, 224), (7, 8, 8, 448)]:
    O, G = 6 * C, 4
    torch.manual_seed(0)
    x = torch.randn(N, H, W, C, device="cuda").to(dt).requires_grad_(True)
    w = torch.nn.Parameter(torch.randn(O, C // G, 3, 3, device="cuda") * (C // G * 9) ** -0.5)
    b = torch.nn.Parameter(torch.zeros(O, device="cuda"))
    g = torch.randn(N, H, W, O, device="cuda").to(dt)
    line = [f"{N}x{H}x{W}, {C} -> {O} / {G} groups:"]
    for dense in (False, True):
        FH.GROUPED_DENSE = dense
        FH.clear_pack_cache()
        def step():
            y = FH.grouped_conv2d(x, w, b, G, N, H, W, ks=3, act=hip.ACT_GELU)
            y.backward(g)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            step()
        e1.record()
        torch.cuda.synchronize()
        line.append(f"{'dense' if dense else 'grouped'} {e0.elapsed_time(e1) / reps * 1e3:7.1f} us")
    print("  ".join(line))
FH.GROUPED_DENSE = True
