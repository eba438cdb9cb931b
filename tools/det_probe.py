import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import cases as C, recipe as R, vmg_oracle as O
from tests.util import build_product, psnr
name = "vmg_reds_few_cfg1"
case = C.CASES[name]; cfg = case["cfg"]
shapes, _ = C.load_fixture(f"tests/golden/{name}.npz")
inp = case["inputs"]()
sd = C.case_state_dict(case, shapes)
with torch.no_grad():
    w = O.vmg_forward({k: v.clone() for k, v in sd.items()}, cfg, inp["x"])
for dt in [torch.bfloat16, torch.float32]:
    outs = []
    for rep in range(4):
        m = build_product(cfg, dt); m.load_state_dict(sd); m.eval()
        with torch.no_grad():
            outs.append(m(inp["x"].cuda()).cpu())
        print(dt, rep, "psnr vs oracle %.2f" % psnr(outs[-1], w), "maxdiff vs rep0 %.3e" % (outs[-1]-outs[0]).abs().max(), flush=True)
