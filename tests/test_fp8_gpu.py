"""fp8 (OCP e4m3, block-scaled) convolution path (SURVEY 8f-4 / BASELINE configs[4] "fp8 MFMA weights"; csrc/conv_fp8.hip).

Two statements per case:
  * EXACTNESS of the kernel against an emulation of its own arithmetic -- activations quantised per (pixel, 32-channel block) and weights per
    output channel by the same rule, then a plain fp32 convolution of the dequantised values: the matrix instruction's products are exact and
    its adder keeps ~14 bits below the largest product, so the two agree to 2e-3 of the output scale (bf16 output rounding included);
  * the stated fp8 TOLERANCE against the unquantised fp32 oracle: relative L2 error of one convolution <= 5e-2 (e4m3 carries 3 mantissa bits:
    2^-4 relative rounding error per element, averaged over a 1 296-term dot product), of a 15-block residual chain <= 3e-2 (the residual
    branch enters with r_scaling = 0.1), whole few-levels model PSNR(fp8 chains, fp32 oracle) >= 38 dB on [0,1] outputs."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _quant_emul(x):
    """torch emulation of the record quantiser: x (..., C) fp32 -> dequantised fp32 of the same shape."""
    C = x.shape[-1]
    nb = (C + 31) // 32
    xp = F.pad(x, (0, nb * 32 - C)).reshape(*x.shape[:-1], nb, 32)
    amax = xp.abs().amax(-1, keepdim=True)
    bits = amax.view(torch.int32)
    sb = ((bits >> 23) & 255) - 8 + ((bits & 0x7FFFFF) > 0x600000).int()
    sb = sb.clamp(1, 254)
    mult = torch.exp2(127.0 - sb.float())
    q = (xp * mult).to(torch.float8_e4m3fn).float() / mult
    return q.reshape(*x.shape[:-1], nb * 32)[..., :C]


def _wquant_emul(w):
    """per-output-channel power-of-two scale, e4m3 values: (O, I, 3, 3) fp32 -> dequantised fp32."""
    amax = w.abs().amax((1, 2, 3), keepdim=True)
    bits = amax.view(torch.int32)
    sb = (((bits >> 23) & 255) - 8 + ((bits & 0x7FFFFF) > 0x600000).int()).clamp(1, 254)
    mult = torch.exp2(127.0 - sb.float())
    return (w * mult).to(torch.float8_e4m3fn).float() / mult


@pytest.mark.parametrize("C", [144, 112])
def test_q8_quantize_matches_the_emulation_bit_for_bit(C):
    from oracle import recipe as R
    from vmg_amd import kernels as K
    x = (R.seeded((2, 9, 21, C), 601) * torch.exp2(R.seeded((2, 9, 21, 1), 602) * 4)).to(torch.bfloat16)  # magnitudes over several octaves
    x[0, 0, 0] = 0  # an all-zero pixel
    rec = K.q8_quantize(x.cuda())
    assert rec.shape[-1] == K.q8_record_bytes(C)
    got = K.q8_dequantize(rec, C).cpu()
    want = _quant_emul(x.float())
    assert torch.equal(got, want)
    nb = (C + 31) // 32
    assert int(rec[..., C:nb * 32].max()) == 0 and int(rec[..., nb * 32 + nb:].max()) == 0  # padding bytes are zero


@pytest.mark.parametrize("C,shape", [(144, (2, 24, 40)), (144, (1, 64, 64)), (112, (3, 19, 37))])
@pytest.mark.parametrize("epi", ["relu", "residual", "lrelu_q8only"])
def test_conv_q8_matches_emulation_and_stated_tolerance(C, shape, epi):
    from oracle import recipe as R
    from vmg_amd import hip
    from vmg_amd import kernels as K
    n, h, w_ = shape
    x = R.seeded((n, h, w_, C), 611).to(torch.bfloat16)
    wt = R.seeded((C, C, 3, 3), 612, (C * 9) ** -0.5)
    b = R.seeded((C,), 613, 0.1)
    res = R.seeded((n, h, w_, C), 614).to(torch.bfloat16)
    pw = K.pack_conv_weight_q8(wt.cuda())
    rec = K.q8_quantize(x.cuda())
    kw = dict(relu=dict(act=hip.ACT_RELU), residual=dict(alpha=0.1, res=res.cuda()), lrelu_q8only=dict(act=hip.ACT_LRELU, slope=0.1, want_bf16=False))[epi]
    out, outq = K.conv_q8_forward(rec, pw, b.cuda(), n, h, w_, **kw)

    def ref(xv, wv):
        y = F.conv2d(xv.permute(0, 3, 1, 2), wv, b, padding=1).permute(0, 2, 3, 1)
        if epi == "relu":
            return F.relu(y)
        if epi == "residual":
            return res.float() + 0.1 * y
        return F.leaky_relu(y, 0.1)
    emul = ref(_quant_emul(x.float()), _wquant_emul(wt))
    exact = ref(x.float(), wt)
    scale = max(1.0, float(exact.abs().max()))
    if out is not None:
        got = out.float().cpu()
        assert float((got - emul.to(torch.bfloat16).float()).abs().max()) <= 2e-3 * scale + 2 ** -8 * scale  # (one bf16 rounding step at most on top)
        rel = float((got - exact).norm() / exact.norm())
        assert rel <= 5e-2, f"relative L2 error of the fp8 convolution vs fp32: {rel:.4f}"
    # the record output: the emulated output (bf16-rounded when a bf16 output exists), quantised by the same rule
    base = emul.to(torch.bfloat16).float() if out is not None else emul
    gotq = K.q8_dequantize(outq, C).cpu()
    wantq = _quant_emul(base)
    # a value that sits on an e4m3 rounding boundary may land on the neighbouring code when the accumulation order differs: bound by one e4m3 step
    step = wantq.abs().clamp_min(2.0 ** -9) * 2.0 ** -3
    assert bool(((gotq - wantq).abs() <= step + 1e-3 * scale).all())
    assert float((gotq - wantq).abs().mean()) <= 2e-3 * scale


def test_conv_q8_data_gradient_pack():
    """transpose_flip: the pack of the data-gradient convolution (input gradient = conv of the output gradient with the transposed, mirrored weights)."""
    from oracle import recipe as R
    from vmg_amd import kernels as K
    C, n, h, w_ = 144, 1, 16, 32
    dy = R.seeded((n, h, w_, C), 621).to(torch.bfloat16)
    wt = R.seeded((C, C, 3, 3), 622, (C * 9) ** -0.5)
    pw = K.pack_conv_weight_q8(wt.cuda(), transpose_flip=True)
    out, _ = K.conv_q8_forward(K.q8_quantize(dy.cuda()), pw, None, n, h, w_, want_q8=False)
    want = F.conv_transpose2d(dy.float().permute(0, 3, 1, 2), wt, padding=1).permute(0, 2, 3, 1)
    rel = float((out.float().cpu() - want).norm() / want.norm())
    assert rel <= 5e-2, rel


@pytest.mark.parametrize("C", [144, 112])
def test_residual_chain_fp8_forward_and_bf16_backward(C):
    """functional.residual_chain with fp8 chains switched on: the forward runs conv1 / conv2 of every block on the fp8 kernel (bf16 residual
    path, bf16 tensors kept for the backward), the backward is the bf16 one.  Output vs the fp32 oracle chain: relative L2 <= 3e-2; input and
    parameter gradients vs the oracle's autograd: cosine >= 0.99 for the input gradient, >= 0.97 per parameter tensor (the bf16 backward works on the
    activations the fp8 forward produced: a block convolution's weight gradient inherits their few-percent error)."""
    import torch.nn as nn
    from oracle import recipe as R, vmg_oracle as O
    from vmg_amd import functional as FH
    from vmg_amd.model import ResidualBlocksWithInputConv
    nblk, n, h, w_ = 15, 2, 32, 32
    m = ResidualBlocksWithInputConv(2 * C, C, nblk, 0.1).cuda()
    sd = {k: R.seeded(tuple(v.shape), 630 + i, (v.shape[1] * 9) ** -0.5 if v.dim() == 4 else 0.05) for i, (k, v) in enumerate(m.state_dict().items())}
    m.load_state_dict(sd)
    a, b = R.seeded((n, h, w_, C), 641).to(torch.bfloat16), R.seeded((n, h, w_, C), 642).to(torch.bfloat16)
    osd = {("r." + k): v.clone().requires_grad_(True) for k, v in sd.items()}
    xo = torch.cat([a, b], -1).float().requires_grad_(True)
    want = O.resblocks(osd, "r.", xo, nblk, 0.1)
    go = R.seeded(tuple(want.shape), 643).to(torch.bfloat16).float()
    wg = torch.autograd.grad(want, [xo] + [osd[k] for k in sorted(osd)], go)
    ad, bd = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    FH.set_fp8_chains(True)
    try:
        got = m([ad, bd])
        got.backward(go.cuda().to(torch.bfloat16))
    finally:
        FH.set_fp8_chains(False)
    assert FH.FP8_STATS["chains"] >= 1
    rel = float((got.float().cpu() - want).norm() / want.norm())
    assert rel <= 3e-2, f"chain output, fp8 vs fp32 oracle: relative L2 {rel:.4f}"
    gx = torch.cat([ad.grad, bd.grad], -1).float().cpu()
    cos = float((gx * wg[0]).sum() / (gx.norm() * wg[0].norm()))
    assert cos >= 0.99, cos
    params = dict(m.named_parameters())
    worst = 1.0
    for k, gw in zip(sorted(osd), wg[1:]):
        g = params[k[2:]].grad.cpu()
        c = float((g * gw).sum() / (g.norm() * gw.norm() + 1e-30))
        worst = min(worst, c)
        assert c >= 0.97, (k, c)
    print(f"fp8 chain C={C}: output rel L2 {rel:.4f}, input-gradient cosine {cos:.4f}, worst parameter-gradient cosine {worst:.4f}")


def test_few_levels_model_with_fp8_chains():
    """VMG-REDS-few_levels forward with the recurrent chains' block convolutions in fp8: PSNR against the fp32 oracle and against the bf16 path."""
    from oracle import cases as C
    from oracle import recipe as R
    from tests.util import build_product, psnr
    case = C.CASES["vmg_reds_few_cfg1"]
    shapes, _ = C.load_fixture(os.path.join(GOLD, "vmg_reds_few_cfg1.npz"))
    sd = C.case_state_dict(case, shapes)
    inp = case["inputs"]()
    outs = {}
    for fp8 in (False, True):
        m = build_product(case["cfg"], torch.bfloat16)
        m.load_state_dict(sd)
        m.eval()
        m.fp8_chains = fp8
        with torch.no_grad():
            outs[fp8] = m(inp["x"].cuda()).float().cpu()
    with torch.no_grad():
        want = case["run"]({k: v.clone() for k, v in sd.items()}, inp)[0]
    tgt = R.synthetic_target(inp["x"])
    p8, p16 = psnr(outs[True], want), psnr(outs[False], want)
    print(f"PSNR vs the fp32 oracle: fp8 chains {p8:.2f} dB, bf16 {p16:.2f} dB; |PSNR(hip, target) - PSNR(oracle, target)| = {abs(psnr(outs[True], tgt) - psnr(want, tgt)):.4f} dB")
    assert p8 >= 38.0, p8
    assert abs(psnr(outs[True], tgt) - psnr(want, tgt)) <= 0.1


def test_few_levels_256x448_forward_bf16_and_fp8_vs_oracle():
    """BASELINE configs[4] at its own frame size and width (the two halves had only been tested apart): the C = 144
    few_levels network on a 3-frame clip of 256 x 448 -- the recurrent chains' launches see M = 2 * 256 * 448 = 229 376 pixels, seven tiles of
    the weight-streaming / fp8 kernel per CU -- in bf16 and with fp8 chains, against the fp32 oracle (about a minute of host time).  Stated
    tolerances: bf16 PSNR >= 40 dB, fp8 chains >= 38 dB, |PSNR(hip, target) - PSNR(oracle, target)| <= 0.05 / 0.1 dB."""
    from oracle import cases as C
    from oracle import recipe as R
    from oracle import vmg_oracle as O
    from tests.util import build_product, psnr
    from vmg_amd import functional as FH
    cfg = C.cfg_reds_few(T=3)
    shapes, _ = C.load_fixture(os.path.join(GOLD, "vmg_reds_few_cfg1.npz"))
    chunk_of, window_of = R.vmg_chunk_lookup(cfg)
    sd = R.recipe_state_dict(shapes, 0, chunk_of, window_of)
    x = R.synthetic_clip(1, 3, 256, 448, 78)
    tgt = R.synthetic_target(x)
    outs = {}
    for fp8 in (False, True):
        m = build_product(cfg, torch.bfloat16)
        m.load_state_dict(sd)
        m.eval()
        m.fp8_chains = fp8
        n0 = FH.FP8_STATS["chains"]
        with torch.no_grad():
            outs[fp8] = m(x.cuda()).float().cpu()
        assert (FH.FP8_STATS["chains"] - n0) == (2 * 3 if fp8 else 0)
        del m
        torch.cuda.empty_cache()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    with torch.no_grad():
        want = O.vmg_forward({k: v.clone() for k, v in sd.items()}, cfg, x)
    p16, p8 = psnr(outs[False], want), psnr(outs[True], want)
    d16, d8 = abs(psnr(outs[False], tgt) - psnr(want, tgt)), abs(psnr(outs[True], tgt) - psnr(want, tgt))
    print(f"256x448, C = 144, T = 3: PSNR vs the fp32 oracle: bf16 {p16:.2f} dB, fp8 chains {p8:.2f} dB; PSNR-vs-target difference {d16:.4f} / {d8:.4f} dB")
    assert p16 >= 40.0 and d16 <= 0.05, (p16, d16)
    assert p8 >= 38.0 and d8 <= 0.1, (p8, d8)
