"""Do two half-batch residual chains on two streams beat one full-batch chain?  The weight-streaming conv has three phases per tile -- an HBM
read burst, ~10 us of MFMAs, an HBM write burst -- and with 256 tiles on 256 CUs every CU is in the same phase at the same time.  Two
kernels of 128 tiles each, out of phase, would overlap one's bursts with the other's MFMAs.   python tools/chain_two_streams.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmg_amd import hip, kernels as K

C, H, W, nblk = 144, 64, 64, 15
torch.manual_seed(0)
dt = torch.bfloat16
w0 = torch.randn(C, 2 * C, 3, 3, device="cuda") * (2 * C * 9) ** -0.5
ws = [torch.randn(C, C, 3, 3, device="cuda") * (C * 9) ** -0.5 for _ in range(2 * nblk)]
bs = [torch.zeros(C, device="cuda") for _ in range(2 * nblk + 1)]
DEEP = int(sys.argv[1]) if len(sys.argv) > 1 else 3  # 3: weight-streaming kernel (one 159 KB workgroup per CU); 2: K-split kernel (three workgroups per CU)
if DEEP == 3:
    pw0 = K.pack_conv_weight_ws(w0, src_ch=[C, C])
    pw1 = [K.pack_conv_weight_ws(w) for w in ws[:nblk]]
    pw2 = [K.pack_conv_weight_ws(w) for w in ws[nblk:]]
else:
    pw0 = K.pack_conv_weight(w0, dt, src_ch=[C, C], cout_tiles=3)
    pw1 = [K.pack_conv_weight(w, dt, src_ch=[C], cout_tiles=3) for w in ws[:nblk]]
    pw2 = [K.pack_conv_weight(w, dt, src_ch=[C], cout_tiles=3) for w in ws[nblk:]]


def chain(x, f):
    return K.resblock_chain_forward([x, f], pw0, bs[0], 0.1, DEEP, pw1, bs[1:nblk + 1], pw2, bs[nblk + 1:], 0.1, DEEP)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


x8, f8 = (torch.randn(8, H, W, C, device="cuda").to(dt) for _ in range(2))
xa, fa, xb, fb = x8[:4].contiguous(), f8[:4].contiguous(), x8[4:].contiguous(), f8[4:].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def one():
    chain(x8, f8)


def two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        chain(xa, fa)
    with torch.cuda.stream(s2):
        chain(xb, fb)
    cur.wait_stream(s1)
    cur.wait_stream(s2)


def halves_serial():
    chain(xa, fa)
    chain(xb, fb)


print("route deep = %d" % DEEP)
print("one chain, 8 frames           : %.1f us (31 convs, M = 32768)" % timed(one))
print("two chains of 4 frames, serial: %.1f us" % timed(halves_serial))
print("two chains of 4 frames, 2 streams: %.1f us" % timed(two))
