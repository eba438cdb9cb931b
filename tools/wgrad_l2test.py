"""Is the 3x3 weight-gradient kernel bound by the memory system or inside the CU?  Same launch geometry twice: 7 pairs x 8 frames of
distinct tensors (132 MB per operand set) vs 56 pairs pointing at ONE frame pair (1.2 MB per operand: every copy hits L2).
Run under rocprofv3 --kernel-trace --stats with an argument 0 (distinct) or 1 (same)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import kernels as K

same = int(sys.argv[1]) if len(sys.argv) > 1 else 0
H = W = 64
C = 144
torch.manual_seed(0)
if same:
    x1 = torch.randn(8, H, W, C, device="cuda").to(torch.bfloat16)
    d1 = torch.randn(8, H, W, C, device="cuda").to(torch.bfloat16)
    x1[1:] = x1[:1]
    d1[1:] = d1[:1]
    # 7 pairs of (8, H, W, C) views that all alias frame 0: expand is not dense, so build pointer-equal tensors by as_strided
    xs = [torch.as_strided(x1, (8, H, W, C), (0, W * C, C, 1)) for _ in range(7)]
    dys = [torch.as_strided(d1, (8, H, W, C), (0, W * C, C, 1)) for _ in range(7)]
else:
    xs = [torch.randn(8, H, W, C, device="cuda").to(torch.bfloat16) for _ in range(7)]
    dys = [torch.randn(8, H, W, C, device="cuda").to(torch.bfloat16) for _ in range(7)]
dw = torch.zeros(C, C, 3, 3, device="cuda")
db = torch.zeros(C, device="cuda")
l = K.hip.lib()
import ctypes
ws = K._wgrad_workspace(dw.device)
xa = (ctypes.c_void_p * 7)(*[t.data_ptr() for t in xs])
da = (ctypes.c_void_p * 7)(*[t.data_ptr() for t in dys])
for _ in range(20):
    K.hip.check(l.vmg_conv_wgrad_batched_ws(1, 3, 7, xa, da, 8, H, W, C, C, C, C, dw.data_ptr(), C, 0, 0, db.data_ptr(), 1.0, ws.data_ptr(), ws.numel(),
                                            K.hip.stream_ptr()), "wgrad")
torch.cuda.synchronize()
print("done", same)
