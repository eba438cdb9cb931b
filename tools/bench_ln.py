"""LayerNorm forward / backward (+ skip-gradient add) at the bench shape (28 x 64 x 64 rows of 144 channels), bf16, stream-event timing.
With the diagnostics build: VMG_LN_BLOCKS=n sets the backward grid, VMG_LN_DBG=1 drops its parameter-gradient tail.  python tools/bench_ln.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import kernels as K

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(0)
M, C = 28 * 64 * 64, 144
dt = torch.bfloat16
x = torch.randn(M, C, device="cuda").to(dt)
dy = torch.randn(M, C, device="cuda").to(dt)
add = torch.randn(M, C, device="cuda").to(dt)
w, b = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")


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


y, mean, rstd = K.layernorm_forward(x, w, b, 1e-5)
into = (torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"))
tf = timed(lambda: K.layernorm_forward(x, w, b, 1e-5))
tb = timed(lambda: K.layernorm_backward(dy, x, mean, rstd, w, into=into))
ta = timed(lambda: K.layernorm_backward(dy, x, mean, rstd, w, into=into, add=add))
dys = [torch.randn(M, C, device="cuda").to(dt) for _ in range(5)]
t5 = timed(lambda: K.layernorm_backward(dys, x, mean, rstd, w, into=into, add=add))
mb = M * C * 2 / 1e6
print("forward %6.1f us (%.0f GB/s)   backward %6.1f us (%.0f GB/s)   backward + add %6.1f us (%.0f GB/s)" %
      (tf, 2 * mb / tf * 1e3, tb, 3 * mb / tb * 1e3, ta, 4 * mb / ta * 1e3), flush=True)
print("backward with five output gradients + add %6.1f us (%.0f GB/s)" % (t5, 8 * mb / t5 * 1e3), flush=True)
