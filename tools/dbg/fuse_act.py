import sys; sys.path.insert(0, "/root/repo")
import torch
from vmg_amd import functional as FH, hip, kernels as K
for dtype in (torch.float32, torch.bfloat16):
    ks, act, slope = 3, hip.ACT_RELU, 0.0
    N, H, W, C0, C1, C2 = 2, 12, 10, 16, 32, 16
    g = torch.Generator(device="cuda").manual_seed(61)
    x0 = torch.randn((N, H, W, C0), generator=g, device="cuda").to(dtype)
    w1 = (torch.randn((C1, C0, ks, ks), generator=g, device="cuda") * (C0 * ks * ks) ** -0.5)
    b1 = torch.randn(C1, generator=g, device="cuda") * 0.1
    w2 = (torch.randn((C2, C1, ks, ks), generator=g, device="cuda") * (C1 * ks * ks) ** -0.5)
    b2 = torch.randn(C2, generator=g, device="cuda") * 0.1
    go = torch.randn((N, H, W, C2), generator=g, device="cuda").to(dtype)
    def run(fuse):
        x = x0.clone().requires_grad_(True)
        ps = [t.clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
        y = FH.conv2d([x], ps[0], ps[1], N, H, W, ks=ks, act=act, slope=slope)
        z = FH.conv2d([y], ps[2], ps[3], N, H, W, ks=ks, fuse_src_act=fuse)
        z.backward(go)
        return [x.grad] + [p.grad for p in ps]
    a, b = run(False), run(True)
    c = run(False)
    for n, u, v, w in zip(("x", "w1", "b1", "w2", "b2"), a, b, c):
        print(dtype, n, float((u.float() - v.float()).abs().max()), "repeat-unfused:", float((u.float() - w.float()).abs().max()), float(u.float().abs().max()))
