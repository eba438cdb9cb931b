import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from vmg_amd import hip, kernels as K
torch.manual_seed(0)
dt = torch.bfloat16
for (N,H,W,Ci,Co,mt) in [(1,16,16,64,64,1),(1,16,16,64,64,2),(1,64,64,64,64,2),(5,256,256,64,64,2),(5,256,256,64,64,1),(5,256,256,144,144,2),(5,256,256,64,144,2),(5,256,256,128,64,2)]:
    x = torch.randn(N,H,W,Ci, device="cuda").to(dt); w = (torch.randn(Co,Ci,3,3, device="cuda")*(Ci*9)**-0.5)
    pw = K.pack_conv_weight(w, dt)
    ref = F.conv2d(x.float().permute(0,3,1,2), w.to(dt).float(), None, padding=1).permute(0,2,3,1)
    o1 = K.conv_forward([x], pw, None, N,H,W, mt=mt)[0].float()
    o2 = K.conv_forward([x], pw, None, N,H,W, mt=mt)[0].float()
    err = (o1-ref).abs()
    bad = (err > 0.05).nonzero()
    print((N,H,W,Ci,Co,mt), "maxerr %.3e" % err.max().item(), "nbad", bad.shape[0], "repeat-equal", torch.equal(o1,o2), flush=True)
    if bad.shape[0]:
        print("  first bad idx", bad[:5].tolist(), " bad pix y%8 hist", torch.bincount(bad[:,1]%8, minlength=8).tolist(), "x%16 hist", torch.bincount(bad[:,2]%16, minlength=16).tolist(), "ch hist/16", torch.bincount(bad[:,3]//16).tolist())
