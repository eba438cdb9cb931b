"""Micro-benchmark of the conv kernel variants.  Python launch overhead (~20 us) hides kernels shorter than that, so
run it under rocprofv3 --kernel-trace --stats and read the kernel durations:  bench_conv.py N [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import hip, kernels as K


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    torch.manual_seed(0)
    dtype = torch.bfloat16
    H = W = 64
    for (Ci, Co) in [(144, 144), (288, 144), (144, 288)]:
        x = torch.randn(N, H, W, Ci, device="cuda").to(dtype)
        r = torch.randn(N, H, W, Co, device="cuda").to(dtype)
        w = torch.randn(Co, Ci, 3, 3, device="cuda") * (Ci * 9) ** -0.5
        b = torch.randn(Co, device="cuda")
        out = torch.empty(N, H, W, Co, device="cuda", dtype=dtype)
        pk = K.pack_conv_weight(w, dtype, cout_tiles=3)
        pw = K.pack_conv_weight_ws(w, cout_tiles=9)
        for name, p, deep in (("ksplit", pk, 2), ("ws", pw, 3)):
            for _ in range(reps):
                K.conv_forward([x], p, b, N, H, W, act=hip.ACT_RELU, out=out, deep=deep)
            for _ in range(reps):
                K.conv_forward([x], p, b, N, H, W, alpha=0.1, res=r, out=out, deep=deep)
            torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
