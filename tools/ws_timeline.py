"""In-kernel timeline of the weight-streaming conv kernel (DIAGNOSTICS build: VMG_DIAG=1 python -m vmg_amd.build, then
VMG_HIP_LIB=vmg_amd/libvmg_hip_diag.so python tools/ws_timeline.py).  Prints, for a consumer wave and a loader wave of the
median workgroup, the 100-MHz stamps relative to the workgroup's first stamp (microseconds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import hip, kernels as K

N, H, W, C = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), 64, 64, 144
x = torch.randn(N, H, W, C, device="cuda").to(torch.bfloat16)
w = torch.randn(C, C, 3, 3, device="cuda") * (C * 9) ** -0.5
b = torch.randn(C, device="cuda")
pw = K.pack_conv_weight_ws(w)
nwg = N * (H // 8) * (W // 16)
buf = torch.zeros(nwg * 8 * 32, dtype=torch.int64, device="cuda")
for _ in range(5):
    K.conv_forward([x], pw, b, N, H, W, act=hip.ACT_RELU)
hip.check(hip.lib().vmg_conv_debug_stamps(buf.data_ptr()), "stamps")
K.conv_forward([x], pw, b, N, H, W, act=hip.ACT_RELU)
torch.cuda.synchronize()
hip.lib().vmg_conv_debug_stamps(None)
st = buf.cpu().reshape(nwg, 8, 32).double() / 100.0  # us
t0 = st[:, :, 0][st[:, :, 0] > 0].min()
start = st[:, 0, 0] - t0
end = st[:, 0, 31] - t0
print("workgroup start (us): min %.2f median %.2f max %.2f; end: min %.2f median %.2f max %.2f" % (start.min(), start.median(), start.max(), end.min(), end.median(), end.max()))
wg = int(torch.argsort(end)[nwg // 2])
for wave, name in ((0, "consumer 0"), (3, "consumer 3"), (4, "loader 0")):
    r = st[wg, wave]
    base = st[wg, :, 0][st[wg, :, 0] > 0].min()
    print(name, " ".join("%d:%.2f" % (i, r[i] - base) for i in range(32) if r[i] > 0))
