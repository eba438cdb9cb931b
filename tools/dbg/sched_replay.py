"""debug: why do SPyNet's parameters move under replay while their learning rate is 0?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import cases as C
from tests.util import build_product
from vmg_amd.train import TrainStep
from vmg_amd.data import synthetic_clip, synthetic_target
cfg = C.cfg_tiny_few(3, is_train=False)
shapes, _ = C.load_fixture("tests/golden/vmg_tiny_few.npz")
sd = C.case_state_dict(C.CASES["vmg_tiny_few"], shapes)
x = synthetic_clip(1, 3, 64, 64, seed=75, device="cuda")
y = synthetic_target(x)
sched = dict(T_period=[40], eta_min=1e-7, flow_fix=6, pre_lr_ratio=0.125, warmup_iter=3)
for graph in (True, False):
    m = build_product(cfg, torch.float32)
    m.load_state_dict(sd)
    m.train()
    ts = TrainStep(m, lr=2e-4, schedule=dict(sched))
    spy0 = {k: p.detach().clone() for k, p in m.spynet.named_parameters()}
    if graph:
        ts.capture(x, y, warmup=2)
    print("graph" if graph else "eager", "iter after capture", ts.iter, "t", ts.opt.t)
    for n in range(9 if not graph else 6):
        loss = ts(x, y)
        torch.cuda.synchronize()
        ch = [(k, float((p.detach() - spy0[k]).abs().max())) for k, p in m.spynet.named_parameters() if not torch.equal(p.detach(), spy0[k])]
        print(" step -> iter", ts.iter, "loss", float(loss), "lrs", [g["lr"] for g in ts.opt.param_groups], "hyper", ts.opt.hyper.cpu().tolist()[0],
              "changed", len(ch), ch[:2], "gnorm spynet", float(torch.cat([p.grad.reshape(-1) for p in m.spynet.parameters()]).norm()))
