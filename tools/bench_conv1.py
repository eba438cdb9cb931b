import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import hip, kernels as K
N, tiles, mt, deep = [int(v) for v in sys.argv[1:5]]
dtype = torch.bfloat16
x = torch.randn(N, 64, 64, 144, device="cuda").to(dtype); w = torch.randn(144, 144, 3, 3, device="cuda") * 0.03; b = torch.randn(144, device="cuda")
out = torch.empty(N, 64, 64, 144, device="cuda", dtype=dtype)
pw = K.pack_conv_weight(w, dtype, cout_tiles=tiles)
for _ in range(30):
    K.conv_forward([x], pw, b, N, 64, 64, act=hip.ACT_RELU, out=out, mt=mt, deep=deep)
torch.cuda.synchronize()
