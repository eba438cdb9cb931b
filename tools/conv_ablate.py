"""Ablation sweep of the conv kernels (VMG_CONV_DBG bits: 1 no weight DMA, 4 no halo staging, 8 no store, 16 return at once,
32 no main loop).  Run under rocprofv3 --kernel-trace and read tools/kstats.py: launches are grouped by a marker kernel
count, so each configuration uses a distinct number of reps:  conv_ablate.py N deep"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import hip, kernels as K

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
deep = int(sys.argv[2]) if len(sys.argv) > 2 else 2
H = W = 64
x = torch.randn(N, H, W, 144, device="cuda").to(torch.bfloat16)
w = torch.randn(144, 144, 3, 3, device="cuda") * 0.03
b = torch.randn(144, device="cuda")
res = torch.randn(N, H, W, 144, device="cuda").to(torch.bfloat16)
out = torch.empty_like(x)
pw = K.pack_conv_weight(w, torch.bfloat16, cout_tiles=5)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for dbg in (0, 32, 4, 8, 4 | 32, 8 | 32, 4 | 8, 4 | 8 | 32, 16):
    os.environ["VMG_CONV_DBG"] = str(dbg)
    for _ in range(5):
        K.conv_forward([x], pw, b, N, H, W, act=hip.ACT_RELU, res=res, out=out, deep=deep)
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(200):
        K.conv_forward([x], pw, b, N, H, W, act=hip.ACT_RELU, res=res, out=out, deep=deep)
    ev[1].record()
    torch.cuda.synchronize()
    print(f"N={N} deep={deep} dbg={dbg:3d}: {ev[0].elapsed_time(ev[1]) / 200 * 1e3:8.2f} us per launch (back-to-back, incl. launch gaps)", flush=True)
