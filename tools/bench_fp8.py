"""The fp8 convolution (csrc/conv_fp8.hip) against the bf16 weight-streaming kernel at the recurrence's shapes, timed with stream events.
python tools/bench_fp8.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vmg_amd import functional as FH, hip, kernels as K  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
C = 144
for (n, h, w) in ((8, 64, 64), (14, 128, 128), (2, 256, 448)):
    M = n * h * w
    x = torch.randn(n, h, w, C, device="cuda").to(torch.bfloat16)
    res = torch.randn(n, h, w, C, device="cuda").to(torch.bfloat16)
    wt = (torch.randn(C, C, 3, 3, device="cuda") * (C * 9) ** -0.5)
    b = torch.randn(C, device="cuda") * 0.1
    pw8 = K.pack_conv_weight_q8(wt)
    rec = K.q8_quantize(x)
    pw = FH.packed(wt, torch.bfloat16, "fwd", [C], tiles=9, deep=3)

    def timeit(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    flop = 2.0 * M * C * C * 9
    t16 = timeit(lambda: K.conv_forward([x], pw, b, n, h, w, alpha=0.1, res=res, deep=3))
    t8a = timeit(lambda: K.conv_q8_forward(rec, pw8, b, n, h, w, alpha=0.1, res=res, want_bf16=True, want_q8=True))
    t8b = timeit(lambda: K.conv_q8_forward(rec, pw8, b, n, h, w, act=hip.ACT_RELU, want_bf16=False, want_q8=True))
    print("M = %7d: bf16 ws %.1f us (%.0f TFLOP/s) | fp8 -> bf16 + records %.1f us (%.0f TFLOP/s) | fp8 -> records only %.1f us (%.0f TFLOP/s)"
          % (M, t16, flop / t16 / 1e6, t8a, flop / t8a / 1e6, t8b, flop / t8b / 1e6))
