"""Times conv_ws_kernel on the dominant shape (8 x 64 x 64 pixels, 144 -> 144, residual + ReLU epilogue) with whatever library VMG_HIP_LIB names: see
tools/ws_lds_ablate.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import hip, kernels as K

N, H, W = 8, 64, 64
x = torch.randn(N, H, W, 144, device="cuda").to(torch.bfloat16)
w = torch.randn(144, 144, 3, 3, device="cuda") * 0.03
b = torch.randn(144, device="cuda")
res = torch.randn(N, H, W, 144, device="cuda").to(torch.bfloat16)
out = torch.empty_like(x)
pw = K.pack_conv_weight_ws(w, cout_tiles=9)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for rnd in range(3):
    for dbg in (0,):
        os.environ["VMG_CONV_DBG"] = str(dbg)
        for _ in range(5):
            K.conv_forward([x], pw, b, N, H, W, act=hip.ACT_RELU, res=res, out=out, deep=3)
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(300):
            K.conv_forward([x], pw, b, N, H, W, act=hip.ACT_RELU, res=res, out=out, deep=3)
        ev[1].record()
        torch.cuda.synchronize()
        print(f"dbg={dbg:4d}: {ev[0].elapsed_time(ev[1]) / 300 * 1e3:8.2f} us per launch (back-to-back)", flush=True)
