"""Squeeze-excite MLP kernels (vmg_se_mlp_fwd / _bwd) at the bench shapes: CALayer.conv_du (28 rows, 144 -> 36 -> 144, ReLU, sigmoid) and the
MorphFC re-weighting (4 rows, 144 -> 36 -> 432, GELU, softmax over triples), fp32, stream-event timing.  python tools/bench_se_mlp.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import hip, kernels as K

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
torch.manual_seed(0)


def timed(fn):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (G, C, Hd, Co, act, mode) in ((28, 144, 36, 144, hip.ACT_RELU, 0), (4, 144, 36, 432, hip.ACT_GELU, 1), (28, 144, 36, 432, hip.ACT_GELU, 1)):
    m = torch.randn(G, C, device="cuda")
    w1, b1 = torch.randn(Hd, C, device="cuda") * C ** -0.5, torch.randn(Hd, device="cuda")
    w2, b2 = torch.randn(Co, Hd, device="cuda") * Hd ** -0.5, torch.randn(Co, device="cuda")
    dout = torch.randn(G, Co, device="cuda")
    pre, out = K.se_mlp_forward(m, w1, b1, w2, b2, act, mode)
    tf = timed(lambda: K.se_mlp_forward(m, w1, b1, w2, b2, act, mode))
    tb = timed(lambda: K.se_mlp_backward(dout, out, m, pre, w1, w2, act, mode, 0.25))
    print("G %2d  %d -> %d -> %d : forward %6.1f us   backward (rows + params) %6.1f us" % (G, C, Hd, Co, tf, tb), flush=True)
