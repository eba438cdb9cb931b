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


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 12, 40, 144, 144, 3), (1, 9, 70, 64, 3, 3), (2, 16, 16, 288, 144, 3), (1, 1, 1000, 144, 288, 1),
                                   (1, 1, 777, 576, 144, 1), (1, 20, 20, 8, 144, 3)])
def test_conv_wgrad(dtype, shape):
    """dW, db accumulate (+=) scale * autograd gradients of the oracle conv."""
    hip, K, O, R = _setup()
    N, H, W, Ci, Co, ks = shape
    x = R.seeded((N, H, W, Ci), 21)
    dy = R.seeded((N, H, W, Co), 22)
    w = torch.zeros(Co, Ci, ks, ks, requires_grad=True)
    b = torch.zeros(Co, requires_grad=True)
    y = O.conv_nhwc(_q(x, dtype), w, b, ks // 2)
    gw, gb = torch.autograd.grad(y, (w, b), _q(dy, dtype))
    init_w, init_b = R.seeded((Co, Ci, ks, ks), 23), R.seeded((Co,), 24)
    dW, db = init_w.clone().cuda(), init_b.clone().cuda()
    K.conv_wgrad(x.cuda().to(dtype), dy.cuda().to(dtype), dW, db, ks, N, H, W, scale=0.5)
    scale = max(1.0, float(gw.abs().max()))
    tol = 2e-4 if dtype == torch.float32 else 2e-3  # bf16 inputs are exact on both sides; fp32 accumulation order differs
    err = float((dW.cpu() - (init_w + 0.5 * gw)).abs().max())
    assert err <= tol * scale, f"dW err {err} scale {scale}"
    errb = float((db.cpu() - (init_b + 0.5 * gb)).abs().max())
    assert errb <= tol * max(1.0, float(gb.abs().max())), f"db err {errb}"


def test_conv_wgrad_slices():
    """Gradient of a weight slice: W[o0:o0+Co, i0:i0+Ci] of a wider parameter (virtual concat / grouped conv)."""
    hip, K, O, R = _setup()
    N, H, W = 1, 16, 32
    x, dy = R.seeded((N, H, W, 288), 25), R.seeded((N, H, W, 144), 26)
    w = torch.zeros(144, 288, 3, 3, requires_grad=True)
    (gw,) = torch.autograd.grad(O.conv_nhwc(x, w, None, 1), w, dy)
    dW = torch.zeros(144, 288, 3, 3, device="cuda")
    xd, dyd = x.cuda(), dy.cuda()
    K.conv_wgrad(xd[..., :144], dyd, dW, None, 3, N, H, W, i0=0)
    K.conv_wgrad(xd[..., 144:], dyd, dW, None, 3, N, H, W, i0=144)
    assert float((dW.cpu() - gw).abs().max()) <= 2e-4 * max(1.0, float(gw.abs().max()))


@pytest.mark.parametrize("shape", [(5, 256, 256, 64, 64, 0), (5, 256, 256, 128, 64, 0), (5, 256, 256, 64, 64, 1), (28, 64, 64, 144, 144, 1)])
def test_conv_repeatable_under_load(shape):
    """Regression: the LDS-DMA weight ring must be waited for explicitly (hipcc does not drain global_load_lds at
    __syncthreads()).  The race only shows when several workgroups share a CU, i.e. on large grids: results must
    be bitwise repeatable and correct there."""
    hip, K, O, R = _setup()
    N, H, W, Ci, Co, mt = shape
    dt = torch.bfloat16
    x = R.seeded((N, H, W, Ci), 31).cuda().to(dt)
    w = R.seeded((Co, Ci, 3, 3), 32, (Ci * 9) ** -0.5).cuda()
    pw = K.pack_conv_weight(w, dt)
    ref = F.conv2d(x[:1].float().permute(0, 3, 1, 2).cpu(), w.to(dt).float().cpu(), None, padding=1).permute(0, 2, 3, 1)
    first = K.conv_forward([x], pw, None, N, H, W, mt=mt)[0]
    assert float((first[:1].float().cpu() - ref).abs().max()) <= 2e-2 * max(1.0, float(ref.abs().max()))
    for _ in range(10):
        again = K.conv_forward([x], pw, None, N, H, W, mt=mt)[0]
        assert torch.equal(first, again)


@pytest.mark.parametrize("shape", [(2, 12, 40, 144, 144, 3), (1, 9, 70, 64, 64, 2), (1, 16, 32, 288, 144, 1), (2, 8, 8, 144, 288, 4),
                                   (1, 64, 64, 144, 144, 14), (2, 20, 36, 3, 144, 2)])
def test_conv_wgrad_batched_large_tile(shape):
    """bf16 3x3 batched weight gradient = the large-tile kernel (LDS-DMA tiles, slabs + ordered reduction); the result
    must equal the sum over pairs of autograd gradients, accumulate into dW/db, and be bitwise reproducible."""
    hip, K, O, R = _setup()
    N, H, W, Ci, Co, P = shape
    dt = torch.bfloat16
    xs = [R.seeded((N, H, W, Ci), 40 + p) for p in range(P)]
    dys = [R.seeded((N, H, W, Co), 60 + p) for p in range(P)]
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    b = torch.zeros(Co, requires_grad=True)
    gw, gb = torch.zeros_like(w), torch.zeros_like(b)
    for x, dy in zip(xs, dys):
        g1, g2 = torch.autograd.grad(O.conv_nhwc(_q(x, dt), w, b, 1), (w, b), _q(dy, dt))
        gw += g1
        gb += g2
    init_w, init_b = R.seeded((Co, Ci, 3, 3), 80), R.seeded((Co,), 81)
    xd, dd = [x.cuda().to(dt) for x in xs], [d.cuda().to(dt) for d in dys]
    if Ci % 8:  # the stem conv: a 3-channel slice of the zero-padded 8-channel input (pixel stride 8)
        xd = [F.pad(x, (0, 8 - Ci % 8))[..., :Ci] for x in xd]
    outs = []
    for rep in range(2):
        dW, db = init_w.clone().cuda(), init_b.clone().cuda()
        K.conv_wgrad_batched(xd, dd, dW, db, 3, N, H, W, scale=0.5)
        outs.append((dW.cpu(), db.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])  # no atomics: reproducible
    scale = max(1.0, float(gw.abs().max()))
    assert float((outs[0][0] - (init_w + 0.5 * gw)).abs().max()) <= 2e-3 * scale
    assert float((outs[0][1] - (init_b + 0.5 * gb)).abs().max()) <= 2e-3 * max(1.0, float(gb.abs().max()))


@pytest.mark.parametrize("tiles", [5, 4])
@pytest.mark.parametrize("shape", [(2, 24, 20, 144, 144), (1, 64, 64, 144, 144), (3, 9, 17, 64, 64), (1, 16, 16, 144, 288), (1, 8, 8, 112, 224),
                                   (2, 16, 32, 64, 3), (1, 10, 10, 8, 16)])
def test_conv3x3_ksplit_variant(shape, tiles):
    """The K-split kernel (deep=2: waves split the K loop, partial tiles reduced through LDS) against the oracle, with
    every fused epilogue: bias + ReLU, activation-gradient mask + residual, scale."""
    hip, K, O, R = _setup()
    dtype = torch.bfloat16
    N, H, W, Ci, Co = shape
    x = R.seeded((N, H, W, Ci), 41)
    w = R.seeded((Co, Ci, 3, 3), 42, (Ci * 9) ** -0.5)
    b = R.seeded((Co,), 43, 0.1)
    conv = O.conv_nhwc(_q(x, dtype), _q(w, dtype), b, 1)
    pw = K.pack_conv_weight(w.cuda(), dtype, cout_tiles=tiles)
    got, _ = K.conv_forward([x.cuda().to(dtype)], pw, b.cuda(), N, H, W, act=hip.ACT_RELU, deep=2)
    _cmp(got, F.relu(conv), dtype, f"k-split conv3x3 {shape}")
    aux, res = R.seeded((N, H, W, Co), 44), R.seeded((N, H, W, Co), 45)
    want = _q(res, dtype) + 0.5 * conv * (_q(aux, dtype) > 0).float()
    got, _ = K.conv_forward([x.cuda().to(dtype)], pw, b.cuda(), N, H, W, alpha=0.5, res=res.cuda().to(dtype), aux=aux.cuda().to(dtype),
                            actgrad=1, deep=2)
    _cmp(got, want, dtype, f"k-split conv3x3 mask+residual {shape}")


def test_conv_ksplit_concat_linear_and_repeatability():
    hip, K, O, R = _setup()
    dtype = torch.bfloat16
    N, H, W, C = 2, 16, 32, 144
    x1, x2 = R.seeded((N, H, W, C), 46), R.seeded((N, H, W, C), 47)
    w = R.seeded((C, 2 * C, 3, 3), 48, (2 * C * 9) ** -0.5)
    b = R.seeded((C,), 49, 0.1)
    want = F.leaky_relu(O.conv_nhwc(torch.cat([_q(x1, dtype), _q(x2, dtype)], -1), _q(w, dtype), b, 1), 0.1)
    pw = K.pack_conv_weight(w.cuda(), dtype, src_ch=[C, C], cout_tiles=5)
    got, _ = K.conv_forward([x1.cuda().to(dtype), x2.cuda().to(dtype)], pw, b.cuda(), N, H, W, act=hip.ACT_LRELU, slope=0.1, deep=2)
    _cmp(got, want, dtype, "k-split concat conv")
    # 1x1 (Linear) with GELU and the pre-activation output, ragged M
    M, Ci, Co = 1000, 144, 288
    x = R.seeded((M, Ci), 50)
    wl = R.seeded((Co, Ci), 51, Ci ** -0.5)
    bl = R.seeded((Co,), 52, 0.1)
    pre = F.linear(_q(x, dtype), _q(wl, dtype), bl)
    pwl = K.pack_conv_weight(wl.cuda(), dtype, cout_tiles=5)
    got, got_pre = K.conv_forward([x.cuda().to(dtype)], pwl, bl.cuda(), 1, 1, M, act=hip.ACT_GELU, want_pre=True, deep=2)
    _cmp(got_pre.reshape(M, Co), pre, dtype, "k-split linear pre")
    _cmp(got.reshape(M, Co), F.gelu(pre), dtype, "k-split linear gelu")
    # run-to-run identical under load (two workgroups per CU, LDS scratch aliased over the halo tile)
    xb = R.seeded((8, 64, 64, C), 53).cuda().to(dtype)
    w2 = R.seeded((C, C, 3, 3), 54, (C * 9) ** -0.5)
    pw2 = K.pack_conv_weight(w2.cuda(), dtype, cout_tiles=5)
    first = K.conv_forward([xb], pw2, None, 8, 64, 64, deep=2)[0].clone()
    ref = K.conv_forward([xb], pw2, None, 8, 64, 64, deep=0)[0]
    assert float((first.float() - ref.float()).abs().max()) <= 2e-2 * max(1.0, float(ref.float().abs().max()))
    for _ in range(20):
        again = K.conv_forward([xb], pw2, None, 8, 64, 64, deep=2)[0]
        assert torch.equal(first, again)


@pytest.mark.parametrize("shape", [(1, 1, 3000, 144, 144, 3), (2, 33, 37, 144, 576, 2), (1, 64, 64, 576, 144, 1), (1, 50, 41, 112, 224, 4),
                                   (4, 64, 64, 432, 144, 2), (1, 1, 2500, 8, 24, 1)])
def test_linear_wgrad_batched_large_tile(shape):
    """bf16 1x1 / Linear batched weight gradient = the GEMM-over-pixels kernel (transposed LDS reads, slabs + ordered
    reduction): sum over pairs of autograd gradients, accumulated into a slice of a wider dW and into db, bitwise
    reproducible; ragged pixel counts and channel counts that are not multiples of the 144-wide tile."""
    hip, K, O, R = _setup()
    N, H, W, Ci, Co, P = shape
    dt = torch.bfloat16
    M = N * H * W
    xs = [R.seeded((M, Ci), 140 + p) for p in range(P)]
    dys = [R.seeded((M, Co), 160 + p) for p in range(P)]
    gw = sum(_q(d, dt).t().double() @ _q(x, dt).double() for x, d in zip(xs, dys)).float()
    gb = sum(_q(d, dt).double().sum(0) for d in dys).float()
    I_total, i0 = Ci + 16, 8
    init_w, init_b = R.seeded((Co, I_total), 180), R.seeded((Co,), 181)
    xd, dd = [x.cuda().to(dt) for x in xs], [d.cuda().to(dt) for d in dys]
    outs = []
    for rep in range(2):
        dW, db = init_w.clone().cuda(), init_b.clone().cuda()
        K.conv_wgrad_batched(xd, dd, dW, db, 1, N, H, W, scale=0.5, i0=i0)
        outs.append((dW.cpu(), db.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    want = init_w.clone()
    want[:, i0:i0 + Ci] += 0.5 * gw
    scale = max(1.0, float(gw.abs().max()))
    assert float((outs[0][0] - want).abs().max()) <= 2e-3 * scale
    assert float((outs[0][1] - (init_b + 0.5 * gb)).abs().max()) <= 2e-3 * max(1.0, float(gb.abs().max()))


def test_conv_ksplit_pixel_shuffle():
    """PixelShuffle(2) + LeakyReLU fused in the K-split kernel's epilogue (the upsampling head, models/vmg.py:629-630)."""
    hip, K, O, R = _setup()
    dtype = torch.bfloat16
    N, H, W, C = 2, 12, 20, 144
    x = R.seeded((N, H, W, C), 14)
    w = R.seeded((4 * C, C, 3, 3), 15, (C * 9) ** -0.5)
    b = R.seeded((4 * C,), 16, 0.1)
    want = F.leaky_relu(O.pixel_shuffle_nhwc(O.conv_nhwc(_q(x, dtype), _q(w, dtype), b, 1)), 0.1)
    pw = K.pack_conv_weight(w.cuda(), dtype, cout_tiles=3)
    got, _ = K.conv_forward([x.cuda().to(dtype)], pw, b.cuda(), N, H, W, act=hip.ACT_LRELU, slope=0.1, pixel_shuffle=True, deep=2)
    assert tuple(got.shape) == (N, 2 * H, 2 * W, C)
    _cmp(got, want, dtype, "k-split pixel shuffle conv")


@pytest.mark.parametrize("tiles", [5, 3])
@pytest.mark.parametrize("shape", [(1000, 144, 144), (114688, 144, 144), (4097, 144, 288), (300, 160, 48), (777, 64, 144), (50, 8, 16), (4113, 288, 144), (70000, 288, 144), (333, 320, 96),
                                   (500, 224, 112)])
def test_linear_wave_autonomous_variant(shape, tiles):
    """The wave-autonomous 1x1 kernel (deep=4: weights resident in registers, 16-row tiles through wave-private LDS) against
    torch: ragged M, several tiles per wave, both block sizes, every 16-byte epilogue, a 288-channel source that the pack splits into two
    blocks (Mlp_cnn.fc2; tiles = 3 only); shapes it does not cover (padded LDS stride at 64 channels) silently take the general kernel.  Results must be identical run to run."""
    hip, K, O, R = _setup()
    dtype = torch.bfloat16
    M, Ci, Co = shape
    x = R.seeded((M, Ci), 71)
    w = R.seeded((Co, Ci), 72, Ci ** -0.5)
    b = R.seeded((Co,), 73, 0.1)
    pre = F.linear(_q(x, dtype), _q(w, dtype), b)
    pw = K.pack_conv_weight(w.cuda(), dtype, cout_tiles=tiles)
    xd = x.cuda().to(dtype)
    got, got_pre = K.conv_forward([xd], pw, b.cuda(), 1, 1, M, act=hip.ACT_GELU, want_pre=True, deep=4)
    _cmp(got_pre.reshape(M, Co), pre, dtype, f"wave-autonomous linear pre {shape}")
    _cmp(got.reshape(M, Co), F.gelu(pre), dtype, f"wave-autonomous linear gelu {shape}")
    aux, res = R.seeded((M, Co), 74), R.seeded((M, Co), 75)
    want = _q(res, dtype) + 0.5 * pre * (_q(aux, dtype) > 0).float()
    kw = dict(alpha=0.5, res=res.cuda().to(dtype).reshape(1, 1, M, Co), aux=aux.cuda().to(dtype).reshape(1, 1, M, Co), actgrad=1, deep=4)
    got, _ = K.conv_forward([xd], pw, b.cuda(), 1, 1, M, **kw)
    _cmp(got.reshape(M, Co), want, dtype, f"wave-autonomous linear mask+residual {shape}")
    for _ in range(3):
        again, _ = K.conv_forward([xd], pw, b.cuda(), 1, 1, M, **kw)
        assert torch.equal(got, again)
    got, _ = K.conv_forward([xd], pw, None, 1, 1, M, act=hip.ACT_RELU, alpha=1.0 / Ci, deep=4)
    _cmp(got.reshape(M, Co), F.relu(F.linear(_q(x, dtype), _q(w, dtype))) / Ci, dtype, f"wave-autonomous linear relu/scale {shape}")


@pytest.mark.parametrize("case", [(2, 40, 52, 64, 64), (1, 8, 16, 64, 64), (3, 19, 37, 64, 64), (1, 64, 300, 8, 64), (2, 33, 20, 32, 48), (1, 24, 48, 48, 16),
                                  (5, 128, 128, 64, 64), (1, 7, 5, 64, 16)])
def test_conv_weights_stationary_variant(case):
    """The weights-stationary 3x3 kernel (deep=6, conv_wstat_kernel: the layer's weights resident in LDS, a persistent workgroup per CU walks
    over 8 x 16-pixel tiles, halo tiles double-buffered by a loader wave) against the oracle conv: ragged H / W (partial tiles at both
    borders), fewer tiles than workgroups, more tiles than workgroups (several per workgroup, both halo buffers), 8..64 input channels, 16..64
    output channels, every 16-byte epilogue.  Results must be identical run to run and identical to the general kernel's."""
    hip, K, O, R = _setup()
    dtype = torch.bfloat16
    N, H, W, Ci, Co = case
    x = R.seeded((N, H, W, Ci), 91)
    w = R.seeded((Co, Ci, 3, 3), 92, (Ci * 9) ** -0.5)
    b = R.seeded((Co,), 93, 0.1)
    pre = O.conv_nhwc(_q(x, dtype), _q(w, dtype), b, 1)
    pw = K.pack_conv_weight(w.cuda(), dtype, cout_tiles=Co // 16)
    xd = x.cuda().to(dtype)
    got, _ = K.conv_forward([xd], pw, b.cuda(), N, H, W, act=hip.ACT_LRELU, slope=0.1, deep=6)
    _cmp(got, F.leaky_relu(pre, 0.1), dtype, f"weights-stationary conv lrelu {case}")
    gen, _ = K.conv_forward([xd], pw, b.cuda(), N, H, W, act=hip.ACT_LRELU, slope=0.1, deep=0)
    assert torch.equal(got, gen), "the weights-stationary kernel must give the general kernel's bits"
    aux, res = R.seeded((N, H, W, Co), 94), R.seeded((N, H, W, Co), 95)
    want = _q(res, dtype) + 0.5 * pre * (_q(aux, dtype) > 0).float()
    kw = dict(alpha=0.5, res=res.cuda().to(dtype), aux=aux.cuda().to(dtype), actgrad=1, deep=6)
    got, _ = K.conv_forward([xd], pw, b.cuda(), N, H, W, **kw)
    _cmp(got, want, dtype, f"weights-stationary conv mask+residual {case}")
    for _ in range(3):
        again, _ = K.conv_forward([xd], pw, b.cuda(), N, H, W, **kw)
        assert torch.equal(got, again)
    got, got_pre = K.conv_forward([xd], pw, None, N, H, W, act=hip.ACT_GELU, want_pre=True, deep=6)
    pre0 = O.conv_nhwc(_q(x, dtype), _q(w, dtype), None, 1)
    _cmp(got_pre, pre0, dtype, f"weights-stationary conv pre {case}")
    _cmp(got, F.gelu(pre0), dtype, f"weights-stationary conv gelu {case}")


# ---------------------------------------------------------------------------------------------- weight-streaming kernel
@pytest.mark.parametrize("shape", [(2, 24, 20, 144, 144), (8, 64, 64, 144, 144), (3, 9, 17, 128, 144), (1, 16, 16, 144, 288), (1, 8, 8, 112, 224),
                                   (2, 20, 33, 224, 112), (1, 12, 16, 448, 112), (1, 30, 50, 32, 144), (2, 20, 36, 144, 256), (1, 9, 17, 64, 128)])
def test_conv_ws_bias_relu(shape):
    """vmg_conv_fwd deep = 3 (128-pixel tiles, loader waves stream the weights through an LDS ring) vs the oracle; shapes cover
    ragged image borders, both channel-block kinds (multiples of 32, and 16 left over), several channel blocks per source
    (halo restage) and several output-channel blocks."""
    hip, K, O, R = _setup()
    dtype = torch.bfloat16
    N, H, W, Ci, Co = shape
    x = R.seeded((N, H, W, Ci), 41)
    w = R.seeded((Co, Ci, 3, 3), 42, (Ci * 9) ** -0.5)
    b = R.seeded((Co,), 43, 0.1)
    want = F.relu(O.conv_nhwc(_q(x, dtype), _q(w, dtype), b, 1))
    tiles = K.ws_eligible(Co, 3, dtype, [Ci])
    assert tiles in (7, 8, 9)
    pw = K.pack_conv_weight_ws(w.cuda(), cout_tiles=tiles)
    got, _ = K.conv_forward([x.cuda().to(dtype)], pw, b.cuda(), N, H, W, act=hip.ACT_RELU, deep=3)
    _cmp(got, want, dtype, f"conv ws {shape}")
    if Co % 48 == 0:
        # round 4: the 48-channel-block instance (three workgroups per tile, two workgroups per CU) -- the same arithmetic in the same order
        # per output element (same K order, fp32 accumulate), so the SAME BITS as the 144-channel-block instance
        pw3 = K.pack_conv_weight_ws(w.cuda(), cout_tiles=3)
        got3, _ = K.conv_forward([x.cuda().to(dtype)], pw3, b.cuda(), N, H, W, act=hip.ACT_RELU, deep=3)
        assert torch.equal(got3, got), f"conv ws, 48-channel blocks {shape}"


def test_conv_ws_concat_residual_gelu_pre_dgrad():
    """The epilogues and operand forms the model uses with the weight-streaming kernel: virtual concat of two sources + LeakyReLU,
    residual + scale, GELU with the pre-activation kept, data gradient with the ReLU mask, and both residual and mask at once."""
    hip, K, O, R = _setup()
    dtype = torch.bfloat16
    N, H, W, C = 2, 16, 32, 144
    x1, x2 = R.seeded((N, H, W, C), 44), R.seeded((N, H, W, C), 45)
    w = R.seeded((C, 2 * C, 3, 3), 46, (2 * C * 9) ** -0.5)
    b = R.seeded((C,), 47, 0.1)
    want = F.leaky_relu(O.conv_nhwc(torch.cat([_q(x1, dtype), _q(x2, dtype)], -1), _q(w, dtype), b, 1), 0.1)
    pw = K.pack_conv_weight_ws(w.cuda(), src_ch=[C, C])
    got, _ = K.conv_forward([x1.cuda().to(dtype), x2.cuda().to(dtype)], pw, b.cuda(), N, H, W, act=hip.ACT_LRELU, slope=0.1)
    _cmp(got, want, dtype, "ws concat conv")
    w2 = R.seeded((C, C, 3, 3), 48, (C * 9) ** -0.5)
    pw2 = K.pack_conv_weight_ws(w2.cuda())
    want2 = _q(x1, dtype) + 0.1 * O.conv_nhwc(_q(x2, dtype), _q(w2, dtype), b, 1)
    got2, _ = K.conv_forward([x2.cuda().to(dtype)], pw2, b.cuda(), N, H, W, alpha=0.1, res=x1.cuda().to(dtype))
    _cmp(got2, want2, dtype, "ws residual conv")
    pre = O.conv_nhwc(_q(x2, dtype), _q(w2, dtype), b, 1)
    got3, got3_pre = K.conv_forward([x2.cuda().to(dtype)], pw2, b.cuda(), N, H, W, act=hip.ACT_GELU, want_pre=True)
    _cmp(got3_pre, pre, dtype, "ws pre-activation")
    _cmp(got3, F.gelu(pre), dtype, "ws gelu")
    # data gradient with the ReLU mask of aux, and with a residual on top
    xg = x2.clone().requires_grad_(True)
    (dx,) = torch.autograd.grad(O.conv_nhwc(xg, _q(w2, dtype), None, 1), xg, _q(x1, dtype))
    aux = R.seeded((N, H, W, C), 49)
    pwd = K.pack_conv_weight_ws(w2.cuda(), transpose_flip=True)
    got4, _ = K.conv_forward([x1.cuda().to(dtype)], pwd, None, N, H, W, aux=aux.cuda().to(dtype), actgrad=1, alpha=0.1)
    _cmp(got4, 0.1 * dx * (_q(aux, dtype) > 0).float(), dtype, "ws dgrad + mask")
    got5, _ = K.conv_forward([x1.cuda().to(dtype)], pwd, None, N, H, W, aux=aux.cuda().to(dtype), actgrad=1, res=x2.cuda().to(dtype))
    _cmp(got5, dx * (_q(aux, dtype) > 0).float() + _q(x2, dtype), dtype, "ws dgrad + mask + residual")


def test_conv_ws_repeatable_and_matches_ksplit():
    """Same inputs, 20 launches back to back on a busy GPU: bit-identical outputs (no race between the loader waves' LDS-DMA and
    the consumer waves' reads), and agreement with the K-split kernel to bf16 rounding."""
    hip, K, O, R = _setup()
    dtype = torch.bfloat16
    N, H, W, C = 8, 64, 64, 144
    x = R.seeded((N, H, W, C), 50).cuda().to(dtype)
    w = R.seeded((C, C, 3, 3), 51, (C * 9) ** -0.5).cuda()
    b = R.seeded((C,), 52, 0.1).cuda()
    pw = K.pack_conv_weight_ws(w)
    pk = K.pack_conv_weight(w, dtype, cout_tiles=3)
    ref, _ = K.conv_forward([x], pk, b, N, H, W, act=hip.ACT_RELU, deep=2)
    first = None
    for i in range(20):
        got, _ = K.conv_forward([x], pw, b, N, H, W, act=hip.ACT_RELU)
        if first is None:
            first = got.clone()
        else:
            assert torch.equal(got, first), f"launch {i} differs from launch 0"
    assert float((first.float() - ref.float()).abs().max()) <= 2e-2 * max(1.0, float(ref.float().abs().max()))


# ---------------------------------------------------------------------------------------------- 7x7 (SPyNet)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(3, 20, 33, 8, 32), (2, 16, 16, 32, 64), (2, 9, 11, 64, 32), (4, 2, 2, 32, 16), (2, 17, 40, 16, 2), (2, 1, 1, 8, 32)])
def test_conv7x7_fwd_dgrad_wgrad(dtype, shape):
    """SPyNetBasicModule's convolutions (models/vmg.py:126-173): forward with ReLU, data gradient (flipped pack) and weight / bias
    gradient of the 7x7 instantiations, incl. 1x1- and 2x2-pixel pyramid levels and the 2-channel flow output."""
    hip, K, O, R = _setup()
    N, H, W, Ci, Co = shape
    x = R.seeded((N, H, W, Ci), 61)
    w = R.seeded((Co, Ci, 7, 7), 62, (Ci * 49) ** -0.5)
    b = R.seeded((Co,), 63, 0.1)
    dy = R.seeded((N, H, W, Co), 64)
    xq = _q(x, dtype).requires_grad_(True)
    wq = _q(w, dtype).requires_grad_(True)
    bq = b.clone().requires_grad_(True)
    pre = O.conv_nhwc(xq, wq, bq, 3)
    gx, gw, gb = torch.autograd.grad(pre, (xq, wq, bq), _q(dy, dtype))
    pw = K.pack_conv_weight(w.cuda(), dtype)
    got, _ = K.conv_forward([x.cuda().to(dtype)], pw, b.cuda(), N, H, W, act=hip.ACT_RELU)
    _cmp(got, F.relu(pre.detach()), dtype, f"conv7x7 {shape}")
    Cop = (Co + 7) // 8 * 8  # the data-gradient's K side (= Co) must be a multiple of 8: zero-padded output channels
    wp = torch.cat([w, w.new_zeros(Cop - Co, Ci, 7, 7)], 0) if Cop != Co else w
    dyp = torch.cat([dy, dy.new_zeros(N, H, W, Cop - Co)], -1) if Cop != Co else dy
    pwd = K.pack_conv_weight(wp.cuda().contiguous(), dtype, transpose_flip=True)
    gotx, _ = K.conv_forward([dyp.cuda().to(dtype).contiguous()], pwd, None, N, H, W)
    _cmp(gotx, gx, dtype, f"conv7x7 dgrad {shape}")
    dW, db = torch.zeros(Co, Ci, 7, 7, device="cuda"), torch.zeros(Co, device="cuda")
    K.conv_wgrad_batched([x.cuda().to(dtype)], [dy.cuda().to(dtype)], dW, db, 7, N, H, W)
    tol = 2e-4 if dtype == torch.float32 else 2e-3
    assert float((dW.cpu() - gw).abs().max()) <= tol * max(1.0, float(gw.abs().max()))
    assert float((db.cpu() - gb).abs().max()) <= tol * max(1.0, float(gb.abs().max()))


@pytest.mark.parametrize("shape", [(6, 40, 70, 32, 64), (5, 33, 64, 64, 32), (48, 8, 8, 16, 2), (7, 21, 45, 8, 32)])
def test_conv7x7_wgrad_slab_kernel(shape):
    """The 7-wave 7x7 weight-gradient kernel with several K slabs, column segments and row blocks (W > 32, H not a multiple of
    the 4-row unit), two (x, dy) pairs, a scale, and accumulation into an existing gradient."""
    hip, K, O, R = _setup()
    N, H, W, Ci, Co = shape
    dt = torch.bfloat16
    xs = [_q(R.seeded((N, H, W, Ci), 71 + i), dt) for i in range(2)]
    dys = [_q(R.seeded((N, H, W, Co), 81 + i), dt) for i in range(2)]
    w = torch.zeros(Co, Ci, 7, 7, requires_grad=True)
    b = torch.zeros(Co, requires_grad=True)
    loss = sum((O.conv_nhwc(x, w, b, 3) * dy).sum() for x, dy in zip(xs, dys))
    gw, gb = torch.autograd.grad(loss, (w, b))
    dW0, db0 = R.seeded((Co, Ci, 7, 7), 91), R.seeded((Co,), 92)
    dW, db = dW0.cuda(), db0.cuda()
    K.conv_wgrad_batched([x.cuda().to(dt) for x in xs], [d.cuda().to(dt) for d in dys], dW, db, 7, N, H, W, scale=0.5)
    wantW, wantb = dW0 + 0.5 * gw, db0 + 0.5 * gb
    assert float((dW.cpu() - wantW).abs().max()) <= 2e-3 * max(1.0, float(wantW.abs().max()))
    assert float((db.cpu() - wantb).abs().max()) <= 2e-3 * max(1.0, float(wantb.abs().max()))
    dW2, db2 = dW0.cuda(), db0.cuda()
    K.conv_wgrad_batched([x.cuda().to(dt) for x in xs], [d.cuda().to(dt) for d in dys], dW2, db2, 7, N, H, W, scale=0.5)
    assert torch.equal(dW, dW2) and torch.equal(db, db2)  # slabs are summed in a fixed order


@pytest.mark.parametrize("shape", [(3, 37, 70, 64, 3), (2, 16, 32, 16, 16), (5, 9, 33, 24, 2), (1, 64, 64, 64, 8)])
def test_conv3x3_small_cout_wgrad_kernel(shape):
    """3x3 weight gradient with <= 16 output channels (conv_last: 64 -> 3) on the three-wave instance of the tap-row kernel: row blocks
    and column segments that do not divide the image, two pairs, a scale, accumulation, and a 3-channel dY that is a slice of a
    zero-padded 8-channel buffer (what the model's backward hands over)."""
    hip, K, O, R = _setup()
    N, H, W, Ci, Co = shape
    dt = torch.bfloat16
    xs = [_q(R.seeded((N, H, W, Ci), 171 + i), dt) for i in range(2)]
    dys = [_q(R.seeded((N, H, W, Co), 181 + i), dt) for i in range(2)]
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    b = torch.zeros(Co, requires_grad=True)
    loss = sum((O.conv_nhwc(x, w, b, 1) * dy).sum() for x, dy in zip(xs, dys))
    gw, gb = torch.autograd.grad(loss, (w, b))
    dW0, db0 = R.seeded((Co, Ci, 3, 3), 191), R.seeded((Co,), 192)

    def padded(d):
        buf = torch.full((N, H, W, (Co + 7) // 8 * 8), 3.0, dtype=dt, device="cuda")  # (the pad channels are computed and dropped)
        buf[..., :Co] = d.cuda().to(dt)
        return buf[..., :Co]

    outs = []
    for _ in range(2):
        dW, db = dW0.cuda(), db0.cuda()
        K.conv_wgrad_batched([x.cuda().to(dt) for x in xs], [padded(d) for d in dys], dW, db, 3, N, H, W, scale=0.5)
        outs.append((dW, db))
    wantW, wantb = dW0 + 0.5 * gw, db0 + 0.5 * gb
    assert float((outs[0][0].cpu() - wantW).abs().max()) <= 2e-3 * max(1.0, float(wantW.abs().max()))
    assert float((outs[0][1].cpu() - wantb).abs().max()) <= 2e-3 * max(1.0, float(wantb.abs().max()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])  # slabs are summed in a fixed order


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", ["none", "relu", "lrelu"])
def test_pixel_unshuffle_actgrad_one_pass(dtype, act):
    """The fused backward of conv -> PixelShuffle -> activation equals depth-to-space undone on both operands followed by the
    activation derivative (bit for bit: same products, same rounding), and torch's pixel_unshuffle on the CPU."""
    hip, K, O, R = _setup()
    N, H, W, c = 2, 5, 7, 24
    code = {"none": hip.ACT_NONE, "relu": hip.ACT_RELU, "lrelu": hip.ACT_LRELU}[act]
    dy = R.seeded((N, 2 * H, 2 * W, c), 301).to(dtype)
    y = R.seeded((N, 2 * H, 2 * W, c), 302).to(dtype)
    got = K.pixel_unshuffle_actgrad(dy.cuda(), y.cuda() if act != "none" else None, N, H, W, code, 0.1, 0.5)
    d = {"none": torch.ones_like(y.float()), "relu": (y.float() > 0).float(), "lrelu": torch.where(y.float() > 0, 1.0, 0.1)}[act]
    want_hi = (dy.float() * d * 0.5).to(dtype)
    want = F.pixel_unshuffle(want_hi.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).contiguous()
    assert tuple(got.shape) == (N, H, W, 4 * c)
    assert torch.equal(got.cpu(), want)
    if act != "none":
        two = K.act_backward(K.pixel_unshuffle(dy.cuda(), N, H, W), K.pixel_unshuffle(y.cuda(), N, H, W), code, 0.1, 0.5)
        assert torch.equal(got, two)


def test_pixel_shuffle_conv_on_the_weight_streaming_kernel():
    """functional.conv2d(pixel_shuffle=True) for a conv the weight-streaming kernel covers (upconv1: 144 -> 576) = that kernel + the vectorised
    depth-to-space pass; output and gradients against the kernel with the fused shuffle store (USE_WS off) and the oracle."""
    hip, K, O, R = _setup()
    from vmg_amd import functional as FH
    N, H, W, C = 2, 16, 24, 144
    dt = torch.bfloat16
    x0 = _q(R.seeded((N, H, W, C), 501), dt)
    w0 = _q(R.seeded((4 * C, C, 3, 3), 502, (9 * C) ** -0.5), dt)
    b0 = R.seeded((4 * C,), 503, 0.1)
    go = _q(R.seeded((N, 2 * H, 2 * W, C), 504), dt)
    want = F.leaky_relu(O.pixel_shuffle_nhwc(O.conv_nhwc(x0, w0, b0, 1)), 0.1)
    res = []
    for ws in (True, False):
        FH.USE_WS = ws
        try:
            x = x0.cuda().to(dt).requires_grad_(True)
            w, b = torch.nn.Parameter(w0.cuda()), torch.nn.Parameter(b0.cuda())
            y = FH.conv2d([x], w, b, N, H, W, ks=3, act=hip.ACT_LRELU, slope=0.1, pixel_shuffle=True)
            g = torch.autograd.grad(y, [x, w, b], go.cuda().to(dt))
            res.append((y.detach(), g))
        finally:
            FH.USE_WS = True
    _cmp(res[0][0], want, dt, "pixel shuffle conv on the ws kernel")
    assert float((res[0][0].float() - res[1][0].float()).abs().max()) <= 2e-2 * max(1.0, float(want.abs().max()))
    for a, b_ in zip(res[0][1], res[1][1]):
        assert float((a.float() - b_.float()).abs().max()) <= 2e-2 * max(1.0, float(b_.float().abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pixel_shuffle_vector_kernel(dtype):
    hip, K, O, R = _setup()
    N, H, W, c = 2, 5, 7, 24
    x = R.seeded((N, H, W, 4 * c), 511).to(dtype)
    got = K.pixel_shuffle(x.cuda(), N, H, W)
    want = F.pixel_shuffle(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).contiguous()
    assert torch.equal(got.cpu(), want)
    assert torch.equal(K.pixel_unshuffle(got, N, H, W).cpu(), x)


def test_repack_all_equals_individual_packs():
    """The one-launch repack plan (vmg_pack_entry / vmg_pack_run) rebuilds exactly the bytes the single pack calls produce: forward and
    data-gradient packs, both layouts, fp32 and bf16, 1x1 / 3x3 / 7x7, a channel slice, after an 'optimizer step' that rewrote the weights."""
    hip, K, O, R = _setup()
    from vmg_amd import functional as FH
    FH.clear_pack_cache()
    ws = [R.seeded((144, 144, 3, 3), 1).cuda(), R.seeded((144, 288, 3, 3), 2).cuda(), R.seeded((64, 32, 7, 7), 3).cuda(),
          R.seeded((576, 144), 4).cuda(), R.seeded((48, 144, 3, 3), 5).cuda()]
    reqs = [(ws[0], torch.bfloat16, "fwd", None, 0, None, 9, 3), (ws[0], torch.bfloat16, "dgrad", None, 0, None, 9, 3),
            (ws[1], torch.bfloat16, "fwd", [144, 144], 0, None, 9, 3), (ws[1], torch.bfloat16, "dgrad", None, 144, 144, 9, 3),
            (ws[2], torch.bfloat16, "fwd", None, 0, None, None, 0), (ws[2], torch.float32, "dgrad", None, 0, None, None, 0),
            (ws[3], torch.bfloat16, "fwd", None, 0, None, None, 0), (ws[3], torch.float32, "fwd", None, 0, None, None, 0),
            (ws[4], torch.bfloat16, "fwd", None, 0, None, 3, 2)]
    first = [FH.packed(w, dt, kind, src_ch=sc, i0=i0, on=on, tiles=tiles, deep=deep) for (w, dt, kind, sc, i0, on, tiles, deep) in reqs]
    before = [p.buf.clone() for p in first]
    for w in ws:
        w.mul_(0.5).add_(0.01)  # "optimizer step"
    FH.bump_weight_epoch()
    FH.repack_all()
    again = [FH.packed(w, dt, kind, src_ch=sc, i0=i0, on=on, tiles=tiles, deep=deep) for (w, dt, kind, sc, i0, on, tiles, deep) in reqs]
    for a, b in zip(first, again):
        assert a is b  # cache hit: repack_all refreshed the entries in place
    FH.clear_pack_cache()
    fresh = [FH.packed(w, dt, kind, src_ch=sc, i0=i0, on=on, tiles=tiles, deep=deep) for (w, dt, kind, sc, i0, on, tiles, deep) in reqs]
    for i, (a, f, b0) in enumerate(zip(again, fresh, before)):
        assert torch.equal(a.buf, f.buf), f"pack {i} differs from the individually packed one"
        assert not torch.equal(a.buf, b0), f"pack {i} was not rebuilt"
    FH.clear_pack_cache()


def test_conv_wgrad3_multi_equals_single_launches():
    """Eleven 3x3 weight gradients of one shape (3 pairs each, own scale, some without bias) through vmg_conv_wgrad3_multi (8 + 3 per launch)
    against the one-problem-per-launch path on the same bf16 operands; both accumulate into an existing gradient."""
    hip, K, O, R = _setup()
    N, H, W, Ci, Co, P = 2, 24, 40, 144, 144, 3
    dt = torch.bfloat16
    probs, single = [], []
    for i in range(11):
        xs = [R.seeded((N, H, W, Ci), 100 + 10 * i + p).cuda().to(dt) for p in range(P)]
        dys = [R.seeded((N, H, W, Co), 500 + 10 * i + p).cuda().to(dt) for p in range(P)]
        dW0, db0 = R.seeded((Co, Ci, 3, 3), 900 + i).cuda(), (R.seeded((Co,), 950 + i).cuda() if i % 3 else None)
        scale = 0.1 if i % 2 else 1.0
        probs.append((xs, dys, dW0.clone(), db0.clone() if db0 is not None else None, scale))
        dW1, db1 = dW0.clone(), (db0.clone() if db0 is not None else None)
        K.conv_wgrad_batched(xs, dys, dW1, db1, 3, N, H, W, scale=scale)
        single.append((dW1, db1))
    K.conv_wgrad3_multi(probs, N, H, W)
    for i, ((_, _, dW, db, _), (dW1, db1)) in enumerate(zip(probs, single)):
        tol = 1e-4 * max(1.0, float(dW1.abs().max()))  # different K splits: fp32 summation order differs
        assert float((dW - dW1).abs().max()) <= tol, f"problem {i}"
        if db is not None:
            assert float((db - db1).abs().max()) <= 1e-4 * max(1.0, float(db1.abs().max())), f"problem {i} bias"


def test_linear_wgrad2_multi_equals_single_launches():
    """Ten Linear weight gradients of one shape (2 pairs each, own scale, some without bias) through vmg_linear_wgrad2_multi (8 + 2 per launch)
    against the one-problem-per-launch path."""
    hip, K, O, R = _setup()
    M, Ci, Co, P = 5000, 144, 288, 2
    dt = torch.bfloat16
    probs, single = [], []
    for i in range(10):
        xs = [R.seeded((M, Ci), 100 + 10 * i + p).cuda().to(dt) for p in range(P)]
        dys = [R.seeded((M, Co), 500 + 10 * i + p).cuda().to(dt) for p in range(P)]
        dW0, db0 = R.seeded((Co, Ci), 900 + i).cuda(), (R.seeded((Co,), 950 + i).cuda() if i % 3 else None)
        scale = 0.25 if i % 2 else 1.0
        probs.append((xs, dys, dW0.clone(), db0.clone() if db0 is not None else None, scale))
        dW1, db1 = dW0.clone(), (db0.clone() if db0 is not None else None)
        K.conv_wgrad_batched(xs, dys, dW1, db1, 1, 1, 1, M, scale=scale)
        single.append((dW1, db1))
    K.linear_wgrad2_multi(probs, M)
    for i, ((_, _, dW, db, _), (dW1, db1)) in enumerate(zip(probs, single)):
        assert float((dW - dW1).abs().max()) <= 1e-4 * max(1.0, float(dW1.abs().max())), f"problem {i}"
        if db is not None:
            assert float((db - db1).abs().max()) <= 1e-4 * max(1.0, float(db1.abs().max())), f"problem {i} bias"


@pytest.mark.parametrize("shape", [(2, 12, 40, 144, 144, 3), (1, 9, 70, 64, 64, 2), (1, 16, 16, 144, 144, 5), (2, 8, 8, 144, 288, 4), (1, 64, 64, 144, 144, 14),
                                   (1, 5, 33, 112, 112, 2), (3, 1, 64, 48, 144, 1), (2, 20, 36, 3, 144, 2)])
def test_conv_wgrad3_kernel_variants_give_the_same_bits(shape):
    """Round 4: conv_wgrad3b_kernel (tile copies as buffer loads with constant per-lane offsets, out-of-image lanes zeroed by the descriptor's
    range check, the copies issued between the MFMA columns) against conv_wgrad3_kernel (per-lane 64-bit pointers and a zero buffer): the same
    tiles in the same order into the same slabs -- the gradients must be equal bit for bit, image borders, partial last segments (W = 40, 70, 33,
    36, 16, 8), partial channel blocks and the 3-channel stem slice included."""
    hip, K, O, R = _setup()
    N, H, W, Ci, Co, P = shape
    dt = torch.bfloat16
    xd = [R.seeded((N, H, W, Ci), 140 + p).cuda().to(dt) for p in range(P)]
    dd = [R.seeded((N, H, W, Co), 160 + p).cuda().to(dt) for p in range(P)]
    if Ci % 8:
        xd = [F.pad(x, (0, 8 - Ci % 8))[..., :Ci] for x in xd]
    outs = []
    lib = hip.lib()
    prev = lib.vmg_conv_wgrad3_variant(-1)
    try:
        for variant in (0, 1):
            lib.vmg_conv_wgrad3_variant(variant)
            dW, db = torch.zeros(Co, Ci, 3, 3, device="cuda"), torch.zeros(Co, device="cuda")
            K.conv_wgrad_batched(xd, dd, dW, db, 3, N, H, W)
            outs.append((dW.cpu(), db.cpu()))
    finally:
        lib.vmg_conv_wgrad3_variant(prev)
    assert prev == 1
    assert torch.isfinite(outs[1][0]).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("geom", [(2, 16, 16, 112, 672, 4), (7, 64, 64, 112, 672, 4), (7, 32, 32, 224, 1344, 4), (7, 8, 8, 448, 2688, 4), (1, 9, 11, 64, 128, 2)])
def test_grouped_conv_dense_route_matches_grouped_launches(geom):
    """Round 4: a grouped 3x3 convolution (Mlp_cnn.fc1 of the full configuration, models/function.py:50-79, n_groups = 4) as ONE launch on the
    dense block-diagonal pack the pack kernels build from the grouped parameter, forward and data gradient, against the G launches of round 3
    (functional.GROUPED_DENSE off).  The extra products are exact zeros; what differs is the fp32 summation order inside a launch: output and
    input gradient within 1e-2 of their scale (bf16 outputs), the weight / bias gradients (same grouped kernel either way, fed the two routes'
    bf16 pre-activation gradients) within 2e-2."""
    hip, K, O, R = _setup()
    from vmg_amd import functional as FH
    N, H, W, C, Oc, G = geom
    dt = torch.bfloat16
    res = []
    for dense in (False, True):
        FH.clear_pack_cache()
        FH.GROUPED_DENSE = dense
        try:
            x = R.seeded((N, H, W, C), 1500).to(dt).cuda().requires_grad_(True)
            w = torch.nn.Parameter(R.seeded((Oc, C // G, 3, 3), 1501, (C // G * 9) ** -0.5).cuda())
            b = torch.nn.Parameter(R.seeded((Oc,), 1502, 0.1).cuda())
            y = FH.grouped_conv2d(x, w, b, G, N, H, W, ks=3, act=hip.ACT_GELU)
            g = R.seeded((N, H, W, Oc), 1503).to(dt).cuda()
            y.backward(g)
            res.append((y.detach().float(), x.grad.float(), w.grad.float(), b.grad.float()))
        finally:
            FH.GROUPED_DENSE = True
    for name, tol, a, bb in zip(("out", "dx", "dW", "db"), (1e-2, 1e-2, 2e-2, 2e-2), res[0], res[1]):
        assert torch.isfinite(bb).all(), name
        err = float((a - bb).abs().max())
        assert err <= tol * max(float(a.abs().max()), 1e-6), f"{name}: {err:.3e} vs scale {float(a.abs().max()):.3e} ({geom})"
    # and against torch's grouped convolution in fp32 on the same bf16-rounded operands
    xr = R.seeded((N, H, W, C), 1500).to(dt).float()
    wr = R.seeded((Oc, C // G, 3, 3), 1501, (C // G * 9) ** -0.5).to(dt).float()
    br = R.seeded((Oc,), 1502, 0.1)
    want = F.gelu(F.conv2d(xr.permute(0, 3, 1, 2), wr, br, padding=1, groups=G)).permute(0, 2, 3, 1)
    assert float((res[1][0].cpu() - want).abs().max()) <= 2e-2 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", ["relu3x3", "gelu1x1", "relu7x7", "lrelu3x3"])
def test_activation_derivative_fused_into_the_consumers_data_gradient(dtype, case):
    """conv2d(..., fuse_src_act=True) (round 4, functional._ActTok): conv -> activation -> conv where the second convolution's data-gradient launch applies the
    first one's activation derivative in its epilogue and the first backward skips its own pass (RCAB's ReLU, the MLP's GELU, SPyNet's 7x7 ReLU chain).  Same
    outputs; every gradient equals the unfused path -- bit-exact for ReLU (a 0 / 1 mask), within a rounding for leaky ReLU / GELU in bf16 -- and act_backward is not
    called on the fused path."""
    from vmg_amd import functional as FH, hip, kernels as K
    ks, act, slope = {"relu3x3": (3, hip.ACT_RELU, 0.0), "gelu1x1": (1, hip.ACT_GELU, 0.0), "relu7x7": (7, hip.ACT_RELU, 0.0), "lrelu3x3": (3, hip.ACT_LRELU, 0.1)}[case]
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
        calls = []
        orig = K.act_backward
        K.act_backward = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            y = FH.conv2d([x], ps[0], ps[1], N, H, W, ks=ks, act=act, slope=slope)
            z = FH.conv2d([y], ps[2], ps[3], N, H, W, ks=ks, fuse_src_act=fuse)
            z.backward(go)
        finally:
            K.act_backward = orig
        return z.detach(), [x.grad] + [p.grad for p in ps], len(calls)

    z0, g0, n0 = run(False)
    z1, g1, n1 = run(True)
    assert n0 == 1 and n1 == 0
    assert torch.equal(z0, z1)
    for k, (a, b) in enumerate(zip(g0, g1)):
        if act == hip.ACT_RELU and (k == 0 or dtype == torch.bfloat16):  # (the fp32 weight gradients are float atomics: run-to-run differences of an ulp)
            assert torch.equal(a, b)
        else:
            tol = 1e-5 if dtype == torch.float32 else 2e-2
            assert float((a.float() - b.float()).abs().max()) <= tol * max(1.0, float(a.float().abs().max()))
