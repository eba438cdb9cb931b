"""MorphFC H / W branch kernel (vmg_morphfc_fwd) at the bench shape (4 x 7 frames of 64 x 64, 144 channels, chunk 8), bf16: forward with and
without the token side output, data-gradient form; stream-event timing.  python tools/bench_morph.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import functional as FH, kernels as K

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(0)
B, T, H, W, C, chunk = 4, 7, 64, 64, 144, 8
dt = torch.bfloat16
x = torch.randn(B, T, H, W, C, device="cuda").to(dt)
dy = torch.randn(B, T, H, W, C, device="cuda").to(dt)
w = torch.randn(C, C, device="cuda") * C ** -0.5
nct = (C + 15) // 16
pf = FH.packed(w, dt, "fwd", [C], tiles=nct)
pd = FH.packed(w, dt, "dgrad", None, 0, C, tiles=nct)


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for axis in ("h", "w"):
    y, _ = K.morphfc_forward(x, axis, chunk, C, pf, None, True, 1.0, 1.0 / C)
    t0 = timed(lambda: K.morphfc_forward(x, axis, chunk, C, pf, None, True, 1.0, 1.0 / C))
    t1 = timed(lambda: K.morphfc_forward(x, axis, chunk, C, pf, None, True, 1.0, 1.0 / C, want_tokens=True))
    t2 = timed(lambda: K.morphfc_forward(dy, axis, chunk, C, pd, None, False, 1.0 / C, 1.0, mask=y, want_tokens=True))
    mb = x.numel() * 2 / 1e6
    print("axis %s: forward %6.1f us (%.0f GB/s over x in + y out)   + tokens %6.1f us   data gradient + tokens %6.1f us" % (axis, t0, 2 * mb / t0 * 1e3, t1, t2), flush=True)
