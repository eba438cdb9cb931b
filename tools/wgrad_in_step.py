"""The batched 3x3 / 1x1 weight-gradient launches of ONE bench train step, each with its problem list and a stream-event time: what the
kernels achieve on the step's own problems (sizes, pairs per problem, problems per launch), not on a synthetic one.
   python tools/wgrad_in_step.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vmg_amd import kernels as K
from vmg_amd.data import synthetic_clip, synthetic_target
from vmg_amd.train import TrainStep

dev = torch.device("cuda", 0)
model = bench.build_model(dev)
ts = TrainStep(model)
x = synthetic_clip(4, 7, 64, 64, seed=1, device=dev)
y = synthetic_target(x)
for _ in range(3):
    ts(x, y)
torch.cuda.synchronize()

log = []
orig3, orig1 = K.conv_wgrad3_multi, K.linear_wgrad2_multi


def wrap3(probs, N, H, W):
    xs0, dys0 = probs[0][0], probs[0][1]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig3(probs, N, H, W)
    e1.record()
    flop = 2.0 * len(probs) * len(xs0) * N * H * W * xs0[0].shape[-1] * dys0[0].shape[-1] * 9
    log.append(("3x3", len(probs), len(xs0), N * H * W, xs0[0].shape[-1], dys0[0].shape[-1], flop, e0, e1))


def wrap1(probs, M):
    xs0, dys0 = probs[0][0], probs[0][1]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig1(probs, M)
    e1.record()
    flop = 2.0 * len(probs) * len(xs0) * M * xs0[0].shape[-1] * dys0[0].shape[-1]
    log.append(("1x1", len(probs), len(xs0), M, xs0[0].shape[-1], dys0[0].shape[-1], flop, e0, e1))


K.conv_wgrad3_multi, K.linear_wgrad2_multi = wrap3, wrap1
from vmg_amd import functional as FH
for mod in (FH,):
    for name in ("conv_wgrad3_multi", "linear_wgrad2_multi"):
        if hasattr(mod, name):
            setattr(mod, name, getattr(K, name))
ts(x, y)
torch.cuda.synchronize()
tot_f = tot_t = 0.0
print("kind  problems pairs   pixels  Cin Cout     GFLOP       us   TFLOP/s   of 2500")
for kind, n, p, M, ci, co, flop, e0, e1 in log:
    us = e0.elapsed_time(e1) * 1e3
    if kind == "3x3":
        tot_f += flop
        tot_t += us
    print(f"{kind}   {n:7d} {p:5d} {M:8d} {ci:4d} {co:4d} {flop / 1e9:9.1f} {us:8.1f} {flop / us / 1e6:9.1f}   {flop / us / 1e6 / 2500:.3f}")
print(f"3x3 total: {tot_f / 1e12:.3f} TFLOP in {tot_t / 1e3:.3f} ms (kernel + reduce pairs by events) = {tot_f / tot_t / 1e6:.1f} TFLOP/s")
