"""The BASELINE configs[4] frame size (256 x 448 LR -> 1024 x 1792 HR) on the HIP path.

  * whole model: the tiny few-levels VMG (C = 16) on one 3-frame clip of 256 x 448, fp32, forward AND every parameter gradient against
    the CPU oracle's autograd (the oracle needs ~20 s for it);
  * single kernels at the HR size: tensors of 0.8 G elements (1.6 GB bf16, 3.3 GB fp32 -- byte offsets beyond 2^31 and 2^32); the LAST
    rows of the LAST frame are compared with the oracle computed on a crop (a 3x3 conv's last 8 output rows need the last 9 input rows),
    which is what 32-bit offset arithmetic anywhere in a kernel would get wrong."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")

N_HR, H_HR, W_HR = 7, 1024, 1792


def test_tiny_model_256x448_forward_and_gradients_fp32():
    from oracle import cases as C
    from oracle import recipe as R
    from tests.test_grad_gpu import _oracle_grads
    from tests.util import build_product, psnr
    case = C.CASES["vmg_tiny_few"]
    cfg = case["cfg"]
    shapes, _ = C.load_fixture(os.path.join(GOLD, "vmg_tiny_few.npz"))
    sd = C.case_state_dict(case, shapes)
    x = R.synthetic_clip(1, 3, 256, 448, 77)
    tgt = R.synthetic_target(x)
    m = build_product(cfg, torch.float32)
    m.load_state_dict(sd)
    m.train()  # (drop-path rates are 0 in this config)
    out = m(x.cuda())
    loss = (out - tgt.cuda()).square().mean()
    loss.backward()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    osd, oloss, oout = _oracle_grads(sd, cfg, x, tgt, want_out=True)
    got = out.detach().cpu()
    assert tuple(got.shape) == (1, 3, 3, 1024, 1792)
    err = float((got - oout).abs().max())
    assert err <= 2e-3, f"256x448 forward: max |hip - oracle| = {err}"
    assert psnr(got, oout) >= 60.0
    assert abs(psnr(got, tgt) - psnr(oout, tgt)) <= 1e-3
    assert abs(float(loss) - oloss) <= 1e-5 * max(1.0, abs(oloss))
    gmax = max(float(v.grad.abs().max()) for v in osd.values() if v.grad is not None)
    for k, p in m.named_parameters():
        want = osd[k].grad
        assert p.grad is not None and want is not None, k
        scale = max(float(want.abs().max()), 1e-3 * gmax)
        e = float((p.grad.cpu() - want).abs().max()) / scale
        assert e <= 5e-3, f"{k}: relative gradient error {e:.3e} at 256x448"


def _tail(t, rows):
    """last `rows` rows of the last frame of a channels-last (n,h,w,c) tensor, fp32 on the CPU."""
    return t[-1, -rows:].float().cpu()


def _conv_crop_oracle(x_tail, w, b, rows_out):
    """3x3 conv (zero padding) of the last frame's bottom rows: x_tail holds the last rows_out + 1 input rows."""
    xt = x_tail.permute(2, 0, 1)[None]                       # (1, c, r+1, w)
    xt = F.pad(xt, (1, 1, 0, 1))                             # zero column padding, zero row below the image; the row above is real data
    y = F.conv2d(xt, w, b)                                   # (1, o, r, w): valid rows
    return y[0].permute(1, 2, 0)[-rows_out:]


@pytest.mark.parametrize("dtype,cin,cout", [(torch.bfloat16, 64, 64), (torch.bfloat16, 64, 3), (torch.float32, 64, 64)])
def test_hr_conv_last_rows(dtype, cin, cout):
    """HRconv (64 -> 64) / conv_last (64 -> 3, 128-pixel tiles) on 7 x 1024 x 1792 pixels."""
    from oracle import recipe as R
    from vmg_amd import functional as FH
    from vmg_amd import hip
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((N_HR, H_HR, W_HR, cin), generator=g, device="cuda", dtype=torch.float32).to(dtype)
    w = R.seeded((cout, cin, 3, 3), 501, (cin * 9) ** -0.5)
    b = R.seeded((cout,), 502, 0.1)
    if dtype == torch.bfloat16:
        w = w.to(dtype).float()
    with torch.no_grad():
        y = FH.conv2d([x], w.cuda(), b.cuda(), N_HR, H_HR, W_HR, ks=3, act=hip.ACT_LRELU, slope=0.1)
    assert tuple(y.shape) == (N_HR, H_HR, W_HR, cout)
    rows = 8
    want = F.leaky_relu(_conv_crop_oracle(_tail(x, rows + 1), w, b, rows), 0.1)
    got = _tail(y, rows)
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
    assert float((got - want).abs().max()) <= tol * max(1.0, float(want.abs().max()))
    # and the first rows of the first frame (the same kernel, offset 0)
    want0 = F.leaky_relu(F.conv2d(F.pad(x[0, :rows + 1].float().cpu().permute(2, 0, 1)[None], (1, 1, 1, 0)), w, b)[0].permute(1, 2, 0)[:rows], 0.1)
    assert float((y[0, :rows].float().cpu() - want0).abs().max()) <= tol * max(1.0, float(want0.abs().max()))


def test_conv_ws_large_pixel_count_last_rows():
    """The weight-streaming kernel (conv3x3 144 -> 144, bf16, + residual) on 8 x 1024 x 1024 pixels: 2.4 GB per tensor."""
    from oracle import recipe as R
    from vmg_amd import functional as FH
    from vmg_amd import hip
    n, h, wd, c = 8, 1024, 1024, 144
    g = torch.Generator(device="cuda").manual_seed(6)
    x = torch.randn((n, h, wd, c), generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    res = torch.randn((n, h, wd, c), generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    w = R.seeded((c, c, 3, 3), 503, (c * 9) ** -0.5).to(torch.bfloat16).float()
    b = R.seeded((c,), 504, 0.1)
    assert FH.choose_tiling(n * h * wd, c, 3, torch.bfloat16, [c])[2] == 3, "this shape must run on the weight-streaming kernel"
    with torch.no_grad():
        y = FH.conv2d([x], w.cuda(), b.cuda(), n, h, wd, ks=3, alpha=0.1, res=res)
    rows = 8
    want = _tail(res, rows) + 0.1 * _conv_crop_oracle(_tail(x, rows + 1), w, b, rows)
    got = _tail(y, rows)
    assert float((got - want).abs().max()) <= 2e-2 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_hr_pixel_shuffle_and_activation_backward_last_rows(dtype):
    """PixelShuffle(2) of (7, 512, 896, 256) -> (7, 1024, 1792, 64), its one-pass backward (un-shuffle + LeakyReLU derivative) and the
    element-wise activation backward on the 0.8 G-element HR map: last rows of the last frame vs torch on the crop."""
    from vmg_amd import kernels as K
    from vmg_amd import hip
    n, h, wd, c4 = N_HR, H_HR // 2, W_HR // 2, 256
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn((n, h, wd, c4), generator=g, device="cuda", dtype=torch.float32).to(dtype)
    y = K.pixel_shuffle(x, n, h, wd)
    assert tuple(y.shape) == (n, 2 * h, 2 * wd, c4 // 4)
    rows = 4
    xt = x[-1, -rows:].float().cpu()                                              # (rows, w, 4c)
    want = F.pixel_shuffle(xt.permute(2, 0, 1)[None], 2)[0].permute(1, 2, 0)       # (2 rows, 2w, c)
    assert torch.equal(y[-1, -2 * rows:].float().cpu(), want)
    # backward of conv -> PixelShuffle -> LeakyReLU(0.1) in one pass: dpre = unshuffle(dy * lrelu'(y))
    dy = torch.randn(y.shape, generator=g, device="cuda", dtype=torch.float32).to(dtype)
    dpre = K.pixel_unshuffle_actgrad(dy, y, n, h, wd, hip.ACT_LRELU, 0.1, 1.0)
    dyt, yt = dy[-1, -2 * rows:].float().cpu(), y[-1, -2 * rows:].float().cpu()
    wd_ = (dyt * torch.where(yt > 0, 1.0, 0.1)).to(dtype).float()
    want_pre = F.pixel_unshuffle(wd_.permute(2, 0, 1)[None], 2)[0].permute(1, 2, 0)
    assert float((dpre[-1, -rows:].float().cpu() - want_pre).abs().max()) <= (1e-2 if dtype == torch.bfloat16 else 1e-6) * max(1.0, float(want_pre.abs().max()))
    # element-wise activation backward over the whole HR map
    ab = K.act_backward(dy, y, hip.ACT_LRELU, 0.1, 1.0)
    assert float((ab[-1, -2 * rows:].float().cpu() - wd_).abs().max()) <= (1e-2 if dtype == torch.bfloat16 else 1e-6) * max(1.0, float(wd_.abs().max()))
