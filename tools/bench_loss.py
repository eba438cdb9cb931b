"""GPU time of the loss (Charbonnier + edge term) forward + backward alone, and of SPyNet forward + backward alone."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vmg_amd
from vmg_amd.train import charbonnier_edge_loss
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda", 0)
x = torch.rand(4, 7, 3, 256, 256, device=dev, requires_grad=True)
y = torch.rand(4, 7, 3, 256, 256, device=dev)
def timeit(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def loss_fb():
    x.grad = None
    charbonnier_edge_loss(x, y).backward()
print("loss fwd+bwd: %.3f ms" % timeit(loss_fb))
spy = vmg_amd.SPyNet(None).to(dev)
a = torch.rand(48, 3, 64, 64, device=dev, requires_grad=True)
b = torch.rand(48, 3, 64, 64, device=dev, requires_grad=True)
def spy_fb():
    spy.zero_grad(); a.grad = None; b.grad = None
    spy(a, b).square().mean().backward()
print("spynet fwd+bwd (48 pairs 64x64): %.3f ms" % timeit(spy_fb))
