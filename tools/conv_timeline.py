"""In-kernel phase timeline of the k-split conv kernel from s_memrealtime stamps (diagnostics):  conv_timeline.py N tiles [channels]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vmg_amd import hip, kernels as K

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H = W = 64
C = int(sys.argv[3]) if len(sys.argv) > 3 else 144
x = torch.randn(N, H, W, C, device="cuda").to(torch.bfloat16)
w = torch.randn(C, C, 3, 3, device="cuda") * 0.03
b = torch.randn(C, device="cuda")
res = torch.randn(N, H, W, C, device="cuda").to(torch.bfloat16)
out = torch.empty_like(x)
pw = K.pack_conv_weight(w, torch.bfloat16, cout_tiles=tiles)
ncb = (C + tiles * 16 - 1) // (tiles * 16)
nwg = N * 16 * 4 * ncb
buf = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device="cuda")
lib = hip.lib()
for _ in range(20):
    K.conv_forward([x], pw, b, N, H, W, act=hip.ACT_RELU, res=res, out=out, deep=2)
torch.cuda.synchronize()
lib.vmg_conv_debug_stamps(buf.data_ptr())
K.conv_forward([x], pw, b, N, H, W, act=hip.ACT_RELU, res=res, out=out, deep=2)
torch.cuda.synchronize()
lib.vmg_conv_debug_stamps(None)
t = buf.cpu().numpy().reshape(nwg, 4, 8).astype(np.float64) * 0.01  # us
t0 = t[:, :, 0].min()
names = ["entry", "bias+prefetch issued", "halo DMA issued", "halo landed", "barrier", "loop done", "barrier+reduce done", "stores issued"]
print(f"N={N} tiles={tiles}: {nwg} workgroups; kernel span {t.max() - t0:.2f} us")
for i, nm in enumerate(names):
    v = t[:, :, i] - t0
    print(f"  [{i}] {nm:20s} abs: min {v.min():6.2f} med {np.median(v):6.2f} max {v.max():6.2f} us" + (f"   | delta from prev: med {np.median(t[:, :, i] - t[:, :, i-1]):6.2f} max {(t[:, :, i] - t[:, :, i-1]).max():6.2f}" if i else ""))
life = t[:, :, 7] - t[:, :, 0]
print(f"  wave lifetime: med {np.median(life):.2f} max {life.max():.2f} us; entry times: first-round waves {np.sum(t[:, :, 0] - t0 < 2.0)} of {nwg * 4}")

first = (t[:, :, 0] - t0) < 3.0
for nm, sel in (("first-round waves", first), ("later waves", ~first)):
    if sel.sum() == 0:
        continue
    d = np.diff(t, axis=2)
    print(f"  {nm} ({int(sel.sum())}): median phase durations " + " ".join(f"{np.median(d[:, :, i][sel]):5.2f}" for i in range(7)))
