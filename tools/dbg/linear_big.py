import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vmg_amd import functional as FH, hip, kernels as K
M = 802816
for cin, cout in ((224, 224), (192, 192), (144, 144), (224, 144)):
    x = torch.randn(M, cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(cout, cin, device="cuda") * cin ** -0.5
    b = torch.randn(cout, device="cuda")
    print(cin, cout, "choose_tiling:", FH.choose_tiling(M, cout, 1, torch.bfloat16, [cin]))
    for tiles, deep, mt in ((None, 0, 1), (None, 0, 2), (None, 1, 1), (None, 1, 2), (3, 4, 1), (5, 4, 1)):
        try:
            pw = FH.packed(w, torch.bfloat16, "fwd", [cin], tiles=tiles, deep=deep)
            for _ in range(3):
                K.conv_forward([x], pw, b, 1, 1, M, act=hip.ACT_RELU, deep=deep, mt=mt)
        except Exception as e:
            print("   tiles", tiles, "deep", deep, "mt", mt, "->", str(e)[:80]); continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            K.conv_forward([x], pw, b, 1, 1, M, act=hip.ACT_RELU, deep=deep, mt=mt)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"   tiles {tiles} deep {deep} mt {mt}: {us:7.1f} us  {(M * (cin + cout) * 2) / us / 1e6:6.2f} TB/s")
