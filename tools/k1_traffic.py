"""The dominant kernel alone (conv3x3 144->144 bf16, M = 32768, weight-streaming kernel, residual epilogue) for the PMC passes behind
bench.py's roofline.traffic and the MFMA-busy figure:
    rocprofv3 --pmc FETCH_SIZE -- python3 tools/k1_traffic.py        (then WRITE_SIZE, then SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ..., each its own pass)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import kernels as K

N, H, W, C = 8, 64, 64, 144
torch.manual_seed(0)
x = torch.randn(N, H, W, C, device="cuda").to(torch.bfloat16)
res = torch.randn(N, H, W, C, device="cuda").to(torch.bfloat16)
w = torch.randn(C, C, 3, 3, device="cuda") * 0.03
b = torch.randn(C, device="cuda")
out = torch.empty_like(x)
pw = K.pack_conv_weight_ws(w, cout_tiles=9)
for _ in range(30):
    K.conv_forward([x], pw, b, N, H, W, alpha=0.1, res=res, out=out)
torch.cuda.synchronize()
print("done")
