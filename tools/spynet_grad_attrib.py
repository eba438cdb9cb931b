"""Where does the bf16 error of SPyNet's parameter gradients enter?  (reds_full measured 0.14 relative L2 on one tensor.)
The flow network's gradient = (flow gradient produced by the main network's backward: flow-warp backward of bf16 features) pushed through
SPyNet's own backward (bf16 7x7 convolutions).  Four runs against the same fp32 oracle gradient separate the two:
    net bf16 / SPyNet bf16   (what tests/test_grad_gpu.py checks)
    net bf16 / SPyNet fp32   -> error that arrives WITH the flow gradient
    net fp32 / SPyNet bf16   -> error of SPyNet's own bf16 forward / backward
    net fp32 / SPyNet fp32   (floor)
python tools/spynet_grad_attrib.py [few_levels|reds_full] [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases as C, recipe as R  # noqa: E402
from tests.test_grad_gpu import _oracle_grads  # noqa: E402
from tests.util import build_product  # noqa: E402
from vmg_amd import functional as FH  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "reds_full"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = C.cfg_reds_few(T=7) if which == "few_levels" else C.cfg_reds_full(T=7)
shapes, _ = C.load_fixture(os.path.join(ROOT, "tests", "golden", "vmg_reds_few_cfg1.npz" if which == "few_levels" else "vmg_reds_full.npz"))
chunk_of, window_of = R.vmg_chunk_lookup(cfg)
sd = R.recipe_state_dict(shapes, 0, chunk_of, window_of)
x = R.synthetic_clip(B, 7, 64, 64, 46)
tgt = R.synthetic_target(x)
torch.set_num_threads(16)
osd, oloss = _oracle_grads(sd, cfg, x, tgt)
spy = [k for k in osd if k.startswith("spynet.") and osd[k].grad is not None]
nmax = max(float(v.grad.norm()) for v in osd.values() if v.grad is not None)
print(f"{which} B={B}: oracle loss {oloss:.6g}; SPyNet tensors with norm >= 1e-3 of the largest: "
      f"{sum(1 for k in spy if float(osd[k].grad.norm()) >= 1e-3 * nmax)} of {len(spy)}")
FH.set_wgrad_mode("deferred")
for net_dt, spy_dt, edge in ((torch.bfloat16, torch.bfloat16, False), (torch.bfloat16, torch.float32, False), (torch.float32, torch.bfloat16, False),
                             (torch.float32, torch.float32, False), (torch.bfloat16, torch.bfloat16, True), (torch.float32, torch.bfloat16, True)):
    m = build_product(cfg, net_dt)
    m.spynet_dtype = spy_dt
    m.spynet.edge_fp32 = edge
    m.load_state_dict(sd)
    m.train()
    out = m(x.cuda())
    loss = (out.float() - tgt.cuda()).square().mean()
    loss.backward()
    FH.flush_deferred_wgrads()
    rows = []
    for k, p in m.named_parameters():
        if not k.startswith("spynet."):
            continue
        w = osd[k].grad.double()
        if float(w.norm()) < 1e-3 * nmax:
            continue
        rows.append((float((p.grad.float().cpu().double() - w).norm() / w.norm()), k))
    rows.sort(reverse=True)
    v = [r[0] for r in rows]
    print(f"net {str(net_dt)[6:]:9s} SPyNet {str(spy_dt)[6:]:9s}{' + fp32 edges' if edge else '            '}: loss {float(loss):.6g}  SPyNet gradient rel L2: max {v[0]:.4f} median {v[len(v) // 2]:.4f}   worst: "
          + ", ".join(f"{k[7:]} {e:.3f}" for e, k in rows[:4]))
    del m, out, loss
