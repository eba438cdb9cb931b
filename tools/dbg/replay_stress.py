"""stress: a captured TrainStep replayed with allocator churn in between; SPyNet (lr 0) must stay bit-identical and finite.  Looks for graph
nodes that write through pointers into memory the caching allocator has handed out again."""
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
bad = 0
for rep in range(6):
    m = build_product(cfg, torch.float32)
    m.load_state_dict(sd)
    m.train()
    ts = TrainStep(m, lr=2e-4, spynet_lr=0.0)
    spy0 = torch.cat([p.detach().reshape(-1).clone() for p in m.spynet.parameters()])
    ts.capture(x, y, warmup=2)
    canaries = []
    for n in range(12):
        loss = ts(x, y)
        torch.cuda.synchronize()
        spy = torch.cat([p.detach().reshape(-1) for p in m.spynet.parameters()])
        ok = torch.equal(spy, spy0)
        # churn: allocations of many sizes, filled with a canary value, checked after the NEXT replay
        for c in canaries:
            if not bool((c == 7.0).all()):
                print(f"rep {rep} step {n}: a canary buffer of {c.numel()} floats was overwritten by the replay")
                bad += 1
        canaries = [torch.full((sz,), 7.0, device="cuda") for sz in (256, 4096, 65536, 1 << 20, 3 << 20) for _ in range(8)]
        if not ok or not torch.isfinite(loss):
            print(f"rep {rep} step {n}: SPyNet changed by {float((spy - spy0).abs().max()):.3e}, loss {float(loss)}")
            bad += 1
    del ts, m
print("stress done, anomalies:", bad)
