"""The fp8 convolution alone (conv3x3 144->144, e4m3 records in, bf16 rows + records out, residual epilogue; M = 32768) for the PMC passes behind
profiles/r03_g_q8_pmc.json:   rocprofv3 --pmc FETCH_SIZE -- python3 tools/q8_traffic.py   (then WRITE_SIZE, then SQ_VALU_MFMA_BUSY_CYCLES ..., each its own pass)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vmg_amd import kernels as K  # noqa: E402

N, H, W, C = 8, 64, 64, 144
torch.manual_seed(0)
x = torch.randn(N, H, W, C, device="cuda").to(torch.bfloat16)
res = torch.randn(N, H, W, C, device="cuda").to(torch.bfloat16)
w = torch.randn(C, C, 3, 3, device="cuda") * 0.03
b = torch.randn(C, device="cuda")
pw = K.pack_conv_weight_q8(w)
rec = K.q8_quantize(x)
out = torch.empty_like(x)
for _ in range(30):
    K.conv_q8_forward(rec, pw, b, N, H, W, alpha=0.1, res=res, want_bf16=True, want_q8=True, out=out)
torch.cuda.synchronize()
print("done")
