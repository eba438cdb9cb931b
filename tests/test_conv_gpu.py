"""HIP implicit-GEMM convolution (through the C-ABI) vs the CPU oracle.  fp32: 1e-4 abs/rel; bf16: inputs and
weights are rounded to bf16 on both sides, so the remaining difference is accumulation order + the final bf16
rounding of the output (tolerance 2e-2 of the output scale)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _setup():
    from vmg_amd import hip, kernels
    from oracle import vmg_oracle as O
    from oracle import recipe as R
    assert torch.cuda.is_available()
    return hip, kernels, O, R


def _cmp(got, want, dtype, what):
    got = got.float().cpu()
    scale = max(1.0, float(want.abs().max()))
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    err = float((got - want).abs().max())
    assert err <= tol * scale, f"{what}: max err {err} (scale {scale}, tol {tol})"


def _q(t, dtype):
    return t.to(dtype).float() if dtype == torch.bfloat16 else t


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 24, 20, 144, 144), (1, 64, 64, 144, 144), (3, 9, 17, 64, 64), (1, 16, 16, 144, 288),
                                   (2, 16, 32, 64, 3), (1, 8, 8, 112, 224)])
def test_conv3x3_bias_relu(dtype, shape):
    hip, K, O, R = _setup()
    N, H, W, Ci, Co = shape
    x = R.seeded((N, H, W, Ci), 1)
    w = R.seeded((Co, Ci, 3, 3), 2, (Ci * 9) ** -0.5)
    b = R.seeded((Co,), 3, 0.1)
    want = F.relu(O.conv_nhwc(_q(x, dtype), _q(w, dtype), b, 1))
    pw = K.pack_conv_weight(w.cuda(), dtype)
    got, _ = K.conv_forward([x.cuda().to(dtype)], pw, b.cuda(), N, H, W, act=hip.ACT_RELU)
    _cmp(got, want, dtype, f"conv3x3 {shape}")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mt", [1, 2])
def test_conv3x3_concat_lrelu_residual(dtype, mt):
    """cat[x1, x2] -> conv 288->144 + LeakyReLU(0.1), then res + 0.1*conv (the recurrent chain's two epilogues)."""
    hip, K, O, R = _setup()
    if dtype == torch.float32 and mt == 2:
        pytest.skip("fp32 always runs MT=1")
    N, H, W, C = 2, 16, 32, 144
    x1, x2 = R.seeded((N, H, W, C), 4), R.seeded((N, H, W, C), 5)
    w = R.seeded((C, 2 * C, 3, 3), 6, (2 * C * 9) ** -0.5)
    b = R.seeded((C,), 7, 0.1)
    want = F.leaky_relu(O.conv_nhwc(torch.cat([_q(x1, dtype), _q(x2, dtype)], -1), _q(w, dtype), b, 1), 0.1)
    pw = K.pack_conv_weight(w.cuda(), dtype, src_ch=[C, C])
    got, _ = K.conv_forward([x1.cuda().to(dtype), x2.cuda().to(dtype)], pw, b.cuda(), N, H, W, act=hip.ACT_LRELU, slope=0.1, mt=mt)
    _cmp(got, want, dtype, "concat conv")
    # residual epilogue
    w2 = R.seeded((C, C, 3, 3), 8, (C * 9) ** -0.5)
    want2 = _q(x1, dtype) + 0.1 * O.conv_nhwc(_q(x2, dtype), _q(w2, dtype), b, 1)
    pw2 = K.pack_conv_weight(w2.cuda(), dtype)
    got2, _ = K.conv_forward([x2.cuda().to(dtype)], pw2, b.cuda(), N, H, W, alpha=0.1, res=x1.cuda().to(dtype), mt=mt)
    _cmp(got2, want2, dtype, "residual conv")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_linear_gelu_and_pre(dtype):
    hip, K, O, R = _setup()
    M, Ci, Co = 1000, 144, 288
    x = R.seeded((M, Ci), 9)
    w = R.seeded((Co, Ci), 10, Ci ** -0.5)
    b = R.seeded((Co,), 11, 0.1)
    pre = F.linear(_q(x, dtype), _q(w, dtype), b)
    pw = K.pack_conv_weight(w.cuda(), dtype)
    got, got_pre = K.conv_forward([x.cuda().to(dtype)], pw, b.cuda(), 1, 1, M, act=hip.ACT_GELU, want_pre=True)
    _cmp(got_pre.reshape(M, Co), pre, dtype, "linear pre")
    _cmp(got.reshape(M, Co), F.gelu(pre), dtype, "linear gelu")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_linear_wide_k_and_slices(dtype):
    """K = 576 (downsample Linear) is split into channel blocks internally; sources may be channel slices."""
    hip, K, O, R = _setup()
    M = 300
    x = R.seeded((M, 576), 12)
    w = R.seeded((144, 576), 13, 576 ** -0.5)
    want = F.linear(_q(x, dtype), _q(w, dtype))
    pw = K.pack_conv_weight(w.cuda(), dtype)
    got, _ = K.conv_forward([x.cuda().to(dtype)], pw, None, 1, 1, M)
    _cmp(got.reshape(M, 144), want, dtype, "K=576 linear")
    # the same as a 3-way virtual concat of slices of one wide tensor
    xd = x.cuda().to(dtype)
    pw3 = K.pack_conv_weight(w.cuda(), dtype, src_ch=[144, 288, 144])
    got3, _ = K.conv_forward([xd[:, :144], xd[:, 144:432], xd[:, 432:]], pw3, None, 1, 1, M)
    _cmp(got3.reshape(M, 144), want, dtype, "sliced sources")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_pixel_shuffle_lrelu(dtype):
    hip, K, O, R = _setup()
    N, H, W, C = 2, 12, 20, 144
    x = R.seeded((N, H, W, C), 14)
    w = R.seeded((4 * C, C, 3, 3), 15, (C * 9) ** -0.5)
    b = R.seeded((4 * C,), 16, 0.1)
    want = F.leaky_relu(O.pixel_shuffle_nhwc(O.conv_nhwc(_q(x, dtype), _q(w, dtype), b, 1)), 0.1)
    pw = K.pack_conv_weight(w.cuda(), dtype)
    got, _ = K.conv_forward([x.cuda().to(dtype)], pw, b.cuda(), N, H, W, act=hip.ACT_LRELU, slope=0.1, pixel_shuffle=True)
    assert tuple(got.shape) == (N, 2 * H, 2 * W, C)
    _cmp(got, want, dtype, "pixel shuffle conv")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_dgrad_pack_and_actgrad(dtype):
    """conv(dY, data-gradient pack) == autograd dX; epilogue mask multiplies by relu'(aux)."""
    hip, K, O, R = _setup()
    N, H, W, Ci, Co = 2, 10, 18, 144, 288
    x = R.seeded((N, H, W, Ci), 17).requires_grad_(True)
    w = R.seeded((Co, Ci, 3, 3), 18, (Ci * 9) ** -0.5)
    dy = R.seeded((N, H, W, Co), 19)
    aux = R.seeded((N, H, W, Ci), 20)
    y = O.conv_nhwc(x, _q(w, dtype), None, 1)
    (dx,) = torch.autograd.grad(y, x, _q(dy, dtype))
    want = dx * (_q(aux, dtype) > 0).float()
    pw = K.pack_conv_weight(w.cuda(), dtype, transpose_flip=True)
    assert pw.cout == Ci and pw.src_ch == [Co]
    got, _ = K.conv_forward([dy.cuda().to(dtype)], pw, None, N, H, W, aux=aux.cuda().to(dtype), actgrad=1)
    _cmp(got, want, dtype, "dgrad")


def test_bad_arguments_raise():
    hip, K, O, R = _setup()
    w = torch.randn(144, 140, 3, 3, device="cuda")
    with pytest.raises(hip.HipError):
        K.pack_conv_weight(w, torch.bfloat16)  # 140 is not a multiple of 8
    pw = K.pack_conv_weight(torch.randn(144, 144, 3, 3, device="cuda"), torch.bfloat16)
    with pytest.raises(hip.HipError):
        K.conv_forward([torch.randn(1, 8, 8, 144, device="cuda")], pw, None, 1, 8, 8)  # fp32 activations vs bf16 pack
    with pytest.raises(hip.HipError):
        K.conv_forward([torch.randn(1, 8, 8, 144)], pw, None, 1, 8, 8)  # CPU tensor: no fallback
