"""stress 2: which gradient goes non-finite first in a replayed step under allocator churn?"""
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
churn = len(sys.argv) < 2 or sys.argv[1] != "nochurn"
for rep in range(3):
    m = build_product(cfg, torch.float32)
    m.load_state_dict(sd)
    m.train()
    ts = TrainStep(m, lr=2e-4, spynet_lr=0.0)
    dbg = torch.zeros_like(ts.opt.g)
    outdbg = {}
    ts.grad_hook = lambda t: dbg.copy_(t.opt.g)
    ts.capture(x, y, warmup=2)
    names = {id(p): k for k, p in m.named_parameters()}
    canaries = []
    ref = None
    for n in range(14):
        loss = ts(x, y)
        torch.cuda.synchronize()
        g = dbg.clone()
        badp = []
        for p, o in zip(ts.opt.params, ts.opt.offsets):
            gg = g[o:o + p.numel()]
            if not torch.isfinite(gg).all():
                badp.append((names[id(p)], int((~torch.isfinite(gg)).sum()), p.numel()))
        pbad = [k for k, p in m.named_parameters() if not torch.isfinite(p).all()]
        print(f"rep {rep} step {n}: loss {float(loss):.6f} non-finite grads {len(badp)} {badp[:6]}  non-finite params {len(pbad)} {pbad[:3]}")
        if badp or pbad:
            break
        if churn:
            canaries = [torch.full((sz,), float("nan"), device="cuda") for sz in (256, 4096, 65536, 1 << 20, 3 << 20) for _ in range(8)]
    del ts, m, dbg
