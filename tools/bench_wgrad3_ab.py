"""A/B of the two 3x3 weight-gradient kernels (vmg_conv_wgrad3_variant 0 / 1) on the chain problem of the bench step: eight 144 -> 144 gradients
per launch, 7 uses of (8, 64, 64) pixels each.  Stream events around `reps` launches.   python tools/bench_wgrad3_ab.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import hip, kernels as K

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N, H, W, C, P, NP = 8, 64, 64, 144, 7, 8
torch.manual_seed(0)
pool = [torch.randn(N, H, W, C, device="cuda").to(torch.bfloat16) for _ in range(2 * P * NP)]
probs = []
for i in range(NP):
    xs, dys = pool[2 * P * i: 2 * P * i + P], pool[2 * P * i + P: 2 * P * (i + 1)]
    probs.append((xs, dys, torch.zeros(C, C, 3, 3, device="cuda"), torch.zeros(C, device="cuda"), 1.0))
lib = hip.lib()
flop = 2.0 * NP * P * N * H * W * C * C * 9
for rnd in range(3):
    for variant in (0, 1):
        lib.vmg_conv_wgrad3_variant(variant)
        for _ in range(3):
            K.conv_wgrad3_multi(probs, N, H, W)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            K.conv_wgrad3_multi(probs, N, H, W)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print(f"variant {variant}: {us:8.1f} us per launch pair (kernel + reduce), {flop / us / 1e6:6.1f} TFLOP/s")
lib.vmg_conv_wgrad3_variant(1)
