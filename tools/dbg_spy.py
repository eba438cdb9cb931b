import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import cases as C, recipe as R
from tests.util import build_product
from vmg_amd import functional as FH
from vmg_amd.train import charbonnier_edge_loss_hip
case = C.CASES["vmg_tiny_few"]
shapes, _ = C.load_fixture("tests/golden/vmg_tiny_few.npz")
sd = C.case_state_dict(case, shapes)
x = R.synthetic_clip(1, 3, 64, 64, 60).cuda(); y = R.synthetic_target(x.cpu()).cuda()
def grads(mode):
    FH.set_wgrad_mode(mode)
    m = build_product(case["cfg"], torch.float32); m.load_state_dict(sd); m.train()
    loss = charbonnier_edge_loss_hip(m(x).float(), y.float()); loss.backward()
    FH.set_wgrad_mode("autograd")
    return {n: p.grad.detach().clone() for n, p in m.named_parameters()}
a = grads("autograd"); b = grads("autograd"); d = grads("deferred")
gmax = max(float(v.abs().max()) for v in a.values())
for n in a:
    if "spynet" in n and "weight" in n:
        s = max(float(a[n].abs().max()), 1e-4 * gmax)
        print(n, "max %.3e" % float(a[n].abs().max()), "auto-auto %.2e" % (float((a[n]-b[n]).abs().max())/s), "auto-deferred %.2e" % (float((a[n]-d[n]).abs().max())/s))
print("gmax", gmax)
# --- TrainStep (FlatAdamW, deferred) on one process vs autograd-mode gradients
from vmg_amd.train import TrainStep
m = build_product(case["cfg"], torch.float32); m.load_state_dict(sd); m.train()
ts = TrainStep(m, lr=1e-4)
cap = {}
ts.grad_hook = lambda t: cap.update({n: p.grad.detach().clone() for n, p in m.named_parameters()})
ts(x, y)
FH.set_wgrad_mode("autograd")
worst = sorted(((float((a[n] - cap[n]).abs().max()) / max(float(a[n].abs().max()), 1e-3 * gmax), n) for n in a), reverse=True)[:6]
print("TrainStep vs autograd:", worst)
