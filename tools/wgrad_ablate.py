"""Ablation of the 3x3 batched weight-gradient kernel (diagnostics build: VMG_DIAG=1 python -m vmg_amd.build, run with
VMG_HIP_LIB=vmg_amd/libvmg_hip_diag.so): VMG_WGRAD_DBG bits 1 no copies after the prologue, 2 no MFMAs, 4 no fragment reads.
With an argument: that setting only (run it under rocprofv3 --kernel-trace --stats: the event timing below is host-bound).
Times the 7-use gradient of a recurrent conv (8 frames of 64x64 per use, 144 -> 144) with stream events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import kernels as K

P, N, H, W, C = 7, 8, 64, 64, 144
torch.manual_seed(0)
xs = [torch.randn(N, H, W, C, device="cuda").to(torch.bfloat16) for _ in range(P)]
dys = [torch.randn(N, H, W, C, device="cuda").to(torch.bfloat16) for _ in range(P)]
dw = torch.zeros(C, C, 3, 3, device="cuda")
db = torch.zeros(C, device="cuda")
for dbg in ([int(sys.argv[1])] if len(sys.argv) > 1 else (0, 1, 2, 4, 6, 3, 7)):
    os.environ["VMG_WGRAD_DBG"] = str(dbg)
    for _ in range(3):
        K.conv_wgrad_batched(xs, dys, dw, db, 3, N, H, W)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        K.conv_wgrad_batched(xs, dys, dw, db, 3, N, H, W)
    e1.record()
    torch.cuda.synchronize()
    print("dbg %d: %.1f us per call (kernel + reduce)" % (dbg, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
