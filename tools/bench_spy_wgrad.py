"""7x7 weight gradients of SPyNet's five convs at every pyramid level (N = 48 frame pairs of the bench batch, bf16), timed with
stream events: python tools/bench_spy_wgrad.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import kernels as K

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = 48
torch.manual_seed(0)
tot = 0.0
for lvl in (64, 32, 16, 8, 4, 2):
    row = []
    for (Ci, Co) in [(8, 32), (32, 64), (64, 32), (32, 16), (16, 2)]:
        x = torch.randn(N, lvl, lvl, Ci, device="cuda").to(torch.bfloat16)
        dy = torch.randn(N, lvl, lvl, Co, device="cuda").to(torch.bfloat16)
        dw = torch.zeros(Co, Ci, 7, 7, device="cuda")
        db = torch.zeros(Co, device="cuda")
        for _ in range(2):
            K.conv_wgrad_batched([x], [dy], dw, db, 7, N, lvl, lvl)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            K.conv_wgrad_batched([x], [dy], dw, db, 7, N, lvl, lvl)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        tot += us
        row.append("%dx%d %7.1f us" % (Ci, Co, us))
    print("level %2d: " % lvl + "   ".join(row), flush=True)
print("sum over levels and convs: %.1f us" % tot)
