"""Micro-benchmark of the conv kernel variants.  Python launch overhead (~20 us) hides kernels shorter than that, so
run it under rocprofv3 --kernel-trace --stats and read the kernel durations:  bench_conv.py N [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import hip, kernels as K


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    torch.manual_seed(0)
    dtype = torch.bfloat16
    H = W = 64
    for (Ci, Co, ks) in [(144, 144, 3), (144, 288, 3), (144, 144, 1)]:
        x = torch.randn(N, H, W, Ci, device="cuda").to(dtype)
        w = torch.randn(Co, Ci, ks, ks, device="cuda") * (Ci * ks * ks) ** -0.5
        b = torch.randn(Co, device="cuda")
        out = torch.empty(N, H, W, Co, device="cuda", dtype=dtype)
        for tiles in (9, 5, 3):
            pw = K.pack_conv_weight(w, dtype, cout_tiles=tiles)
            for mt, deep in ((1, 0), (1, 2)) if tiles <= 5 else ((1, 0),):
                for _ in range(reps):
                    K.conv_forward([x], pw, b, N, H, W, act=hip.ACT_RELU, out=out, mt=mt, deep=deep)
                torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
