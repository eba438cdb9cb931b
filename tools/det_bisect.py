import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import cases as C
from tests.util import build_product
name = "vmg_reds_few_cfg1"
case = C.CASES[name]; cfg = case["cfg"]
shapes, _ = C.load_fixture(f"tests/golden/{name}.npz")
inp = case["inputs"]()
sd = C.case_state_dict(case, shapes)
def run():
    m = build_product(cfg, torch.bfloat16); m.load_state_dict(sd); m.eval()
    rec = []
    def hook(name):
        def f(mod, i, o):
            if isinstance(o, torch.Tensor):
                rec.append((name, o.detach().float().cpu().clone()))
        return f
    for n, mod in m.named_modules():
        if n: mod.register_forward_hook(hook(n))
    with torch.no_grad():
        out = m(inp["x"].cuda())
    return rec
a = run(); b = run()
bad = 0
for (na, ta), (nb, tb) in zip(a, b):
    d = (ta - tb).abs().max().item()
    if d > 0:
        print("MISMATCH", na, tuple(ta.shape), d, flush=True); bad += 1
        if bad > 12: break
print("modules", len(a), "bad", bad)
