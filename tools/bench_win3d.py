"""The 3-D window attention kernels at the train_swin workload's shape -- (4, 8, 32, 32, 144) tokens (7 frames padded to 8), window (4, 8, 8), 8 heads x 18,
unshifted and shifted -- VALU form vs MFMA form (vmg_win3d_variant 0 / 1): stream events around `reps` forward and backward launches.
    python tools/bench_win3d.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import hip, kernels as K

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, D, H, W, C, heads, wt = 4, 8, 32, 32, 144, 8, 4
torch.manual_seed(0)
dt = torch.bfloat16
q = (torch.randn(B, D, H, W, C, device="cuda") * 0.7).to(dt)
kv = (torch.randn(B, D, H, W, 2 * C, device="cuda") * 0.7).to(dt)
bq, bkv = torch.randn(C, device="cuda") * 0.3, torch.randn(2 * C, device="cuda") * 0.3
table = torch.randn((2 * wt - 1) * 225, heads, device="cuda") * 0.5
dout = torch.randn(B, D, H, W, C, device="cuda").to(dt)
lib = hip.lib()
# algorithmic FLOPs: per query 2 * d * (N - 64) for QK^T and the same for PV; backward 2.5x (S, dP, dQ, dK, dV with S recomputed once per pass: 7 contractions)
flop_f = 2.0 * 2 * B * D * H * W * C * (wt * 64 - 64)
for shift in ((0, 0, 0), (2, 4, 4)):
    for variant in (0, 1):
        lib.vmg_win3d_variant(variant)
        out, lse = K.win3d_attn_forward(q, kv, bq, bkv, table, heads, wt, shift)
        for _ in range(3):
            K.win3d_attn_forward(q, kv, bq, bkv, table, heads, wt, shift)
            K.win3d_attn_backward(q, kv, bq, bkv, table, out, lse, dout, heads, wt, shift)
        torch.cuda.synchronize()
        ts = []
        for fn in (lambda: K.win3d_attn_forward(q, kv, bq, bkv, table, heads, wt, shift),
                   lambda: K.win3d_attn_backward(q, kv, bq, bkv, table, out, lse, dout, heads, wt, shift)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
        print(f"shift {shift} variant {variant} ({'MFMA' if variant else 'VALU'}): forward {ts[0]:7.1f} us ({flop_f / ts[0] / 1e6:6.1f} TFLOP/s)   backward {ts[1]:7.1f} us")
lib.vmg_win3d_variant(1)
