"""1x1 / Linear kernels in isolation (read tools/kstats.py on a rocprofv3 kernel trace): bench_linear.py [M] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import hip, kernels as K
M = int(sys.argv[1]) if len(sys.argv) > 1 else 114688
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
torch.manual_seed(0)
x = torch.randn(M, 144, device="cuda").to(torch.bfloat16)
w = torch.randn(144, 144, device="cuda") / 12
b = torch.randn(144, device="cuda")
out = torch.empty(1, 1, M, 144, device="cuda", dtype=torch.bfloat16)
for tiles, deep in ((9, 0), (5, 4), (3, 4)):
    pw = K.pack_conv_weight(w, torch.bfloat16, cout_tiles=tiles)
    for _ in range(reps):
        K.conv_forward([x], pw, b, 1, 1, M, act=hip.ACT_RELU, out=out, deep=deep)
    torch.cuda.synchronize()
# Mlp_cnn.fc2 (288 -> 144 + residual) and its data gradient (144 -> 288): general kernel vs the wave-autonomous one, stream-event timing
def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (ci, co) in ((288, 144), (144, 288)):
    xx = torch.randn(M, ci, device="cuda").to(torch.bfloat16)
    ww = torch.randn(co, ci, device="cuda") / ci ** 0.5
    rr = torch.randn(1, 1, M, co, device="cuda").to(torch.bfloat16)
    oo = torch.empty(1, 1, M, co, device="cuda", dtype=torch.bfloat16)
    for tiles, deep in ((9, 0), (3, 4), (5, 4)):
        pw = K.pack_conv_weight(ww, torch.bfloat16, cout_tiles=tiles)
        t = timed(lambda: K.conv_forward([xx], pw, None, 1, 1, M, res=rr, out=oo, deep=deep))
        print("%d -> %d  tiles %d deep %d : %6.1f us" % (ci, co, tiles, deep, t), flush=True)
print("done")
