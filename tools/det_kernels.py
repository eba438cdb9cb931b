import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import hip, kernels as K, functional as FH
torch.manual_seed(0)
dt = torch.bfloat16
def rep(name, fn, n=30):
    ref = fn().float().clone()
    bad = 0
    for i in range(n):
        # perturb allocator / timing
        junk = torch.randn(1 << (10 + i % 12), device="cuda")
        o = fn().float()
        if not torch.equal(o, ref): bad += 1
    print(f"{name:50s} mismatches {bad}/{n}", flush=True)
for (N,H,W,Ci,Co,ks,ps,mt) in [(5,64,64,144,144,3,False,1),(5,64,64,144,576,3,True,0),(5,128,128,144,256,3,True,0),(5,256,256,64,64,3,False,0),
                             (5,256,256,64,3,3,False,0),(5,64,64,8,144,3,False,0),(1,1,20480,144,144,1,False,0),(1,1,5120,576,144,1,False,0),(1,1,20480,40,144,1,False,0), (1,1,20480,288,144,1,False,0)]:
    x = torch.randn(N,H,W,Ci, device="cuda").to(dt); w = torch.randn(Co,Ci,ks,ks, device="cuda")*(Ci*ks*ks)**-0.5; b = torch.randn(Co, device="cuda")
    pw = K.pack_conv_weight(w, dt)
    rep(f"conv {N}x{H}x{W} {Ci}->{Co} ks{ks} ps{ps}", lambda: K.conv_forward([x], pw, b, N,H,W, act=hip.ACT_LRELU, slope=0.1, pixel_shuffle=ps, mt=mt)[0])
# 2-source
x1 = torch.randn(5,64,64,144, device="cuda").to(dt); x2 = torch.randn(5,64,64,144, device="cuda").to(dt)
w = torch.randn(144,288,3,3, device="cuda")*0.02; pw = K.pack_conv_weight(w, dt, src_ch=[144,144])
rep("conv 2src", lambda: K.conv_forward([x1,x2], pw, None, 5,64,64)[0])
for C in (144, 576, 36):
    x = torch.randn(20480, C, device="cuda").to(dt); g = torch.randn(C, device="cuda"); bb = torch.randn(C, device="cuda")
    rep(f"layernorm C={C}", lambda: K.layernorm_forward(x, g, bb)[0])
# packing determinism
w = torch.randn(144,144,3,3, device="cuda")
rep("pack", lambda: K.pack_conv_weight(w, dt).buf.float())
