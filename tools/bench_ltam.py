"""Trajectory window attention (vmg_ltam_fwd / vmg_ltam_bwd) at the bench batch's shape (8 x 64 x 64 pixels, 144 channels, 4 heads,
2 x 2 windows) for 1..6 key-frames, bf16, timed with stream events (the backward time includes zeroing and casting the fp32
accumulators, as in the step): python tools/bench_ltam.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import kernels as K

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.manual_seed(0)
n, h, w, c, heads = 8, 64, 64, 144, 4  # (both direction sweeps of the 4-clip batch run in lockstep: 8 frames per call)
dt = torch.bfloat16
tot_f = tot_b = 0.0
for t in (1, 2, 3, 4, 5, 6):
    q = torch.randn(n, h, w, c, device="cuda").to(dt)
    keys = [torch.randn(n, h, w, c, device="cuda").to(dt) for _ in range(t)]
    vals = [torch.randn(n, h, w, c, device="cuda").to(dt) for _ in range(t)]
    ys, xs = torch.meshgrid(torch.arange(h, device="cuda", dtype=torch.float32), torch.arange(w, device="cuda", dtype=torch.float32), indexing="ij")
    loc = torch.stack([xs, ys], 0).repeat(t, 1, 1)[None].repeat(n, 1, 1, 1) + 3.0 * torch.randn(n, 2 * t, h, w, device="cuda")
    loc = loc.contiguous()
    rpe = torch.randn(heads, 4, 4, device="cuda")
    decay = torch.rand(heads, device="cuda")
    scale = (c // heads) ** -0.5
    dout = torch.randn(n, h, w, c, device="cuda").to(dt)

    def timed(fn):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    out, lse = K.ltam_forward(q, keys, vals, loc, rpe, decay, heads, 2, 2, scale)
    tf = timed(lambda: K.ltam_forward(q, keys, vals, loc, rpe, decay, heads, 2, 2, scale))
    tb = timed(lambda: K.ltam_backward(q, keys, vals, loc, rpe, decay, out, lse, dout, heads, 2, 2, scale))
    tot_f += tf
    tot_b += tb
    print("t = %d : forward %7.1f us   backward %7.1f us" % (t, tf, tb), flush=True)
print("sum over t = 1..6 (the recurrence of one 7-frame batch, both directions): forward %.1f us, backward %.1f us" % (tot_f, tot_b))
