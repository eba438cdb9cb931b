"""Micro-benchmark of the dominant kernel: conv3x3 C->C on (B, 64, 64) frames (K1 of SURVEY 2.2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import hip, kernels as K


def timeit(fn, iters=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    torch.manual_seed(0)
    for dtype in (torch.bfloat16, torch.float32):
        for (N, H, W, Ci, Co, ks) in [(4, 64, 64, 144, 144, 3), (28, 64, 64, 144, 144, 3), (28, 64, 64, 144, 288, 3),
                                      (28, 64, 64, 144, 576, 3), (28, 64, 64, 144, 144, 1), (28, 64, 64, 288, 144, 1)]:
            x = torch.randn(N, H, W, Ci, device="cuda").to(dtype)
            w = torch.randn(Co, Ci, ks, ks, device="cuda") * (Ci * ks * ks) ** -0.5
            b = torch.randn(Co, device="cuda")
            pw = K.pack_conv_weight(w, dtype)
            out = torch.empty(N, H, W, Co, device="cuda", dtype=dtype)
            flops = 2.0 * N * H * W * Ci * Co * ks * ks
            for mt in ((1, 2) if dtype == torch.bfloat16 else (1,)):
                us = timeit(lambda: K.conv_forward([x], pw, b, N, H, W, act=hip.ACT_RELU, out=out, mt=mt))
                print(f"{str(dtype):15s} N={N:2d} {Ci}->{Co} ks={ks} mt={mt}: {us:8.1f} us  {flops / us / 1e6:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
