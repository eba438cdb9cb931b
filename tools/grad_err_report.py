"""bf16 whole-model gradient error per parameter tensor against the fp32 oracle's autograd (the data behind the stated bf16 gradient
tolerance of tests/test_grad_gpu.py).  python tools/grad_err_report.py [few_levels|reds_full]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases as C, recipe as R  # noqa: E402
from tests.test_grad_gpu import _oracle_grads  # noqa: E402
from tests.util import build_product  # noqa: E402
from vmg_amd import functional as FH  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "few_levels"
cfg = C.cfg_reds_few(T=7) if which == "few_levels" else C.cfg_reds_full(T=7)
shapes, _ = C.load_fixture(os.path.join(ROOT, "tests", "golden", "vmg_reds_few_cfg1.npz" if which == "few_levels" else "vmg_reds_full.npz"))
chunk_of, window_of = R.vmg_chunk_lookup(cfg)
sd = R.recipe_state_dict(shapes, 0, chunk_of, window_of)
x = R.synthetic_clip(1, 7, 64, 64, 46)
tgt = R.synthetic_target(x)
FH.set_wgrad_mode("deferred")
m = build_product(cfg, torch.bfloat16)
m.load_state_dict(sd)
m.train()
out = m(x.cuda())
loss = (out.float() - tgt.cuda()).square().mean()
loss.backward()
torch.set_num_threads(16)
osd, oloss = _oracle_grads(sd, cfg, x, tgt)
rows = []
nmax = max(float(osd[k].grad.norm()) for k, _ in m.named_parameters())
for k, p in m.named_parameters():
    g, w = p.grad.float().cpu().double(), osd[k].grad.double()
    rows.append((float((g - w).norm()) / max(float(w.norm()), 1e-30), float(w.norm()) / nmax, k, p.numel()))
rows.sort(reverse=True)
print(f"{which}: loss {float(loss):.6g} vs {oloss:.6g}")
for r in rows[:25]:
    print("rel L2 %.4f  norm/nmax %.2e  numel %8d  %s" % (r[0], r[1], r[3], r[2]))
for name, sel in (("spynet", lambda k: k.startswith("spynet.")), ("rest", lambda k: not k.startswith("spynet."))):
    v = sorted(r[0] for r in rows if sel(r[2]) and r[1] >= 1e-3)
    print(name, "n", len(v), "median %.4f p90 %.4f max %.4f" % (v[len(v) // 2], v[int(len(v) * 0.9)], v[-1]))
