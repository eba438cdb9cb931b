"""HIP kernels of the trajectory recurrence vs the CPU oracle: flow warp (bilinear/border fwd + bwd, nearest location
advection) and the fused LTAM window attention (fwd + bwd).  fp32: 1e-5 / 1e-4; bf16 (inputs pre-rounded): 2e-2."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mods():
    from oracle import recipe as R
    from oracle import vmg_oracle as O
    from oracle import cases as C
    from vmg_amd import functional as FH
    return R, O, C, FH


def _q(t, dtype):
    return t.to(dtype).float() if dtype == torch.bfloat16 else t


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 16, 24, 144), (1, 9, 7, 32)])
def test_flow_warp_bilinear_border_fwd_bwd(dtype, shape):
    R, O, C, FH = _mods()
    n, h, w, c = shape
    x = _q(R.seeded((n, h, w, c), 51), dtype).requires_grad_(True)
    flow = R.seeded((n, h, w, 2), 52, 3.0).requires_grad_(True)  # incl. far out-of-range samples (border clamp)
    gy = _q(R.seeded((n, h, w, c), 53), dtype)
    want = O.flow_warp(x.permute(0, 3, 1, 2), flow, padding="border").permute(0, 2, 3, 1)
    wdx, wdf = torch.autograd.grad(want, (x, flow), gy)
    xd = x.detach().cuda().to(dtype).requires_grad_(True)
    fd = flow.detach().cuda().requires_grad_(True)
    got = FH.grid_sample_flow(xd, fd, "bilinear", "border")
    gdx, gdf = torch.autograd.grad(got, (xd, fd), gy.cuda().to(dtype))
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert float((got.float().cpu() - want).abs().max()) <= tol * max(1.0, float(want.abs().max()))
    assert float((gdx.float().cpu() - wdx).abs().max()) <= (1e-4 if dtype == torch.float32 else 3e-2) * max(1.0, float(wdx.abs().max()))
    assert float((gdf.cpu() - wdf).abs().max()) <= (1e-3 if dtype == torch.float32 else 3e-2) * max(1.0, float(wdf.abs().max()))


def test_location_advection_matches_exactly():
    """nearest + border on integer-valued location maps must reproduce the oracle bit for bit (index arithmetic)."""
    R, O, C, FH = _mods()
    n, k2, h, w = 2, 6, 20, 28
    loc = C.int_locations(n, k2 // 2, h, w, 54)
    flow = R.seeded((n, h, w, 2), 55, 2.5)
    flow[0, :4, :4] = torch.tensor([0.5, -0.5])  # exact half-pixel offsets: round-half-to-even path
    want = O.flow_warp(loc, flow, mode="nearest", padding="border")
    got = FH.warp_locations(loc.cuda(), flow.cuda()).cpu()
    assert torch.equal(got, want)


def _ltam_oracle(O, sd, q, keys, vals, loc):
    n, h, w, c = q.shape
    return O.ltam_wins(sd, "", q, torch.stack(keys, 1), torch.zeros_like(q), torch.stack(vals, 1), loc, 4, (2, 2))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 3, 16, 16, 144), (1, 1, 10, 12, 32), (1, 5, 8, 8, 16)])
def test_ltam_attention_fwd_bwd(dtype, shape):
    R, O, C, FH = _mods()
    n, t, h, w, c = shape
    q = _q(R.seeded((n, h, w, c), 61), dtype).requires_grad_(True)
    keys = [_q(R.seeded((n, h, w, c), 62 + j), dtype).requires_grad_(True) for j in range(t)]
    vals = [_q(R.seeded((n, h, w, c), 72 + j), dtype).requires_grad_(True) for j in range(t)]
    loc = C.int_locations(n, t, h, w, 82)
    rpe = R.seeded((4, 4, 4), 83, 0.5).requires_grad_(True)
    decay = 1 - 2 ** (-5 - torch.arange(3, -1, -1, dtype=torch.float32))
    sd = {"proj.weight": torch.eye(c), "proj.bias": torch.zeros(c), "relative_pos_encoding": rpe, "decay_v": decay}
    want = _ltam_oracle(O, sd, q, keys, vals, loc)
    go = _q(R.seeded((n, h, w, c), 84), dtype)
    wg = torch.autograd.grad(want, [q, rpe] + keys + vals, go)
    dev = lambda z: z.detach().cuda().to(dtype).requires_grad_(True)
    qd, kd, vd = dev(q), [dev(k) for k in keys], [dev(v) for v in vals]
    rd = rpe.detach().cuda().requires_grad_(True)
    got = FH.ltam_attention(qd, kd, vd, loc.cuda(), rd, decay.cuda(), 4, 2, 2, (c // 4) ** -0.5)
    gg = torch.autograd.grad(got, [qd, rd] + kd + vd, go.cuda().to(dtype))
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert float((got.float().cpu() - want).abs().max()) <= tol * max(1.0, float(want.abs().max()))
    gtol = 2e-4 if dtype == torch.float32 else 4e-2
    names = ["dq", "drpe"] + [f"dk{j}" for j in range(t)] + [f"dv{j}" for j in range(t)]
    for nm, a, b in zip(names, gg, wg):
        err = float((a.float().cpu() - b).abs().max())
        assert err <= gtol * max(1.0, float(b.abs().max())), f"{nm}: {err} vs scale {float(b.abs().max())}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ltam_grad_bank_sums_the_calls_that_share_a_key_frame(dtype):
    """Key / value frames wrapped in functional.grad_bank: three attention calls (1, 2, 3 key-frames, as the recurrence issues them) scatter
    their gradients into one accumulator per frame; the result equals autograd's sum of the per-call gradients, including a frame that
    is also used outside the attention."""
    R, O, C, FH = _mods()
    n, h, w, c = 2, 12, 16, 32
    mk = lambda seed: _q(R.seeded((n, h, w, c), seed), dtype).cuda().to(dtype)
    frames_k, frames_v = [mk(301 + j) for j in range(3)], [mk(311 + j) for j in range(3)]
    qs = [mk(321 + j) for j in range(3)]
    gos = [mk(331 + j) for j in range(3)]
    rpe0 = R.seeded((4, 4, 4), 341, 0.5).cuda()
    decay = (1 - 2 ** (-5 - torch.arange(3, -1, -1, dtype=torch.float32))).cuda()

    def run(banked):
        ks = [f.clone().requires_grad_(True) for f in frames_k]
        vs = [f.clone().requires_grad_(True) for f in frames_v]
        rpe = rpe0.clone().requires_grad_(True)
        kk = [FH.grad_bank(k) if banked else k for k in ks]
        vv = [FH.grad_bank(v) if banked else v for v in vs]
        total = (ks[0] * gos[0]).sum().float()  # a use of frame 0 outside the attention
        for t in (1, 2, 3):
            loc = C.int_locations(n, t, h, w, 350 + t).cuda()
            out = FH.ltam_attention(qs[t - 1], kk[:t], vv[:t], loc, rpe, decay, 4, 2, 2, (c // 4) ** -0.5)
            total = total + (out * gos[t - 1]).sum().float()
        return torch.autograd.grad(total, ks + vs + [rpe])

    a, b = run(True), run(False)
    tol = 1e-5 if dtype == torch.float32 else 3e-2
    for i, (x, y) in enumerate(zip(a, b)):
        assert float((x.float() - y.float()).abs().max()) <= tol * max(1.0, float(y.float().abs().max())), f"gradient {i}"


def test_flow_warp_backward_with_many_clamped_pixels_bf16_vs_fp32():
    """Border padding with large flows: whole rows / columns of output pixels clamp onto the same border pixels, so a source pixel collects
    dozens of contributions.  The bf16 path accumulates them in fp32 and rounds once: its dx must agree with the fp32 path's to bf16 rounding
    (2^-8 of each element), also where the sums are large -- a bf16 running sum (round 2) drifted by several percent there."""
    R, O, C, FH = _mods()
    n, h, w, c = 1, 24, 40, 64
    x = R.seeded((n, h, w, c), 361).to(torch.bfloat16)
    flow = R.seeded((n, h, w, 2), 362, 1.0)
    flow[:, :, : w // 2, 0] -= 60.0   # the left half samples far left of the image: clamped onto column 0
    flow[:, : h // 3, :, 1] -= 40.0   # the top third: clamped onto row 0
    gy = (R.seeded((n, h, w, c), 363).abs() + 0.5).to(torch.bfloat16)  # same-sign gradients: the border sums grow to ~h * w / 6 terms

    def run(dt):
        xd = x.cuda().to(dt).requires_grad_(True)
        fd = flow.cuda().requires_grad_(True)
        out = FH.grid_sample_flow(xd, fd, "bilinear", "border")
        gx, gf = torch.autograd.grad(out, (xd, fd), gy.cuda().to(dt))
        return gx.float().cpu(), gf.cpu()
    g16, f16 = run(torch.bfloat16)
    g32, f32 = run(torch.float32)
    assert float(g32.abs().max()) > 50.0  # the clamped border really collects many contributions
    assert float(((g16 - g32).abs() / (g32.abs() + 1e-3)).max()) <= 2 ** -8 + 1e-3
    assert float((f16 - f32).abs().max()) <= 2e-2 * max(1.0, float(f32.abs().max()))
    # and two bf16 runs agree to the fp32 sum's arrival-order rounding
    g16b, _ = run(torch.bfloat16)
    assert float(((g16 - g16b).abs() / (g16.abs() + 1e-3)).max()) <= 2 ** -7


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,t", [(1, 1), (2, 5), (3, 7)])
def test_pair_frames_is_transpose_flip_cat_and_its_gradient(dtype, n, t):
    """functional.pair_frames (vmg_frame_gather): the (t, 2n) arrangement the lock-step sweeps work on = cat([flip(x^T), x^T], 1), bit-exact;
    its gradient = the autograd gradient of that torch expression (two terms per element, fp32 sum: bit-exact in fp32, one rounding in bf16)."""
    from vmg_amd import functional as FH
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn((n, t, 6, 5, 16), generator=g, device="cuda").to(dtype).requires_grad_(True)
    got = FH.pair_frames(x)
    xt = x.detach().clone().requires_grad_(True)
    xtt = xt.transpose(0, 1)
    want = torch.cat([xtt.flip(0), xtt], 1)
    assert got.shape == want.shape and torch.equal(got, want)
    go = torch.randn(want.shape, generator=g, device="cuda").to(dtype)
    got.backward(go)
    want.float().backward(go.float()) if dtype == torch.float32 else want.backward(go)
    if dtype == torch.float32:
        assert torch.equal(x.grad, xt.grad)
    else:
        ref = (go.float()[:, :n].flip(0) + go.float()[:, n:]).transpose(0, 1)  # the exact sum, rounded once
        assert torch.equal(x.grad, ref.to(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k", [2, 3, 4])
def test_fan_out_sums_the_consumers_gradients_in_one_pass(dtype, k):
    """functional.fan_out (vmg_sum_n, round 4): k handles on one tensor; its gradient is the fp32 sum of the handles' gradients rounded once -- for bf16 at least
    as close to the exact sum as autograd's pairwise bf16 adds; a handle without a gradient is skipped; no-grad mode hands out the tensor itself."""
    from vmg_amd import functional as FH
    g = torch.Generator(device="cuda").manual_seed(41)
    x = torch.randn((2, 6, 5, 16), generator=g, device="cuda").to(dtype).requires_grad_(True)
    hs = FH.fan_out(x, k)
    assert len(hs) == k and all(torch.equal(h, x) for h in hs)
    gos = [torch.randn(x.shape, generator=g, device="cuda").to(dtype) for _ in range(k)]
    torch.autograd.backward(list(hs), gos)
    exact = torch.stack([t.double() for t in gos]).sum(0)
    if dtype == torch.float32:
        assert torch.allclose(x.grad.double(), exact, rtol=0, atol=1e-5)
    else:
        pair = gos[0]
        for t in gos[1:]:
            pair = pair + t
        assert float((x.grad.double() - exact).abs().max()) <= float((pair.double() - exact).abs().max()) + 1e-12
        assert torch.equal(x.grad, exact.float().to(dtype)) or float((x.grad.float() - exact.float().to(dtype).float()).abs().max()) <= 2 ** -8 * float(exact.abs().max())
    x.grad = None
    hs = FH.fan_out(x, 3)
    (hs[0] * 2).sum().backward()  # two handles unused
    assert torch.equal(x.grad, torch.full_like(x, 2.0))
    with torch.no_grad():
        assert all(h is x for h in FH.fan_out(x, 3))


@pytest.mark.parametrize("shape,r", [((2, 3, 2, 30, 26), 4), ((1, 2, 2, 64, 64), 4), ((1, 1, 2, 9, 7), 4), ((2, 1, 2, 5, 5), 3), ((1, 1, 2, 8, 8), 1)])
def test_flow_smoothing_kernel_fwd_bwd(shape, r):
    """functional.flow_smooth (vmg_flow_smooth, round 4) against the torch spelling it replaces -- F.pad(reflect) + adaptive_avg_pool2d + nearest x r + crop
    (models/function.py:1466-1478) -- forward and gradient, sizes with and without padding on either axis."""
    import numpy as np
    import torch.nn.functional as F
    from vmg_amd import functional as FH
    g = torch.Generator(device="cuda").manual_seed(51)
    flow = torch.randn(shape, generator=g, device="cuda").requires_grad_(True)
    got = FH.flow_smooth(flow, r)
    fr = flow.detach().clone().requires_grad_(True)
    B, T, C, H, W = shape
    hf, wf = int(np.ceil(H / r)) * r, int(np.ceil(W / r)) * r
    f = F.pad(fr.reshape(-1, C, H, W), (0, wf - W, 0, hf - H), mode="reflect")
    f = F.adaptive_avg_pool2d(f, (hf // r, wf // r))
    want = F.interpolate(f, scale_factor=r, mode="nearest")[..., :H, :W].reshape(shape)
    assert got.shape == want.shape and torch.allclose(got, want, rtol=1e-6, atol=1e-6)
    go = torch.randn(shape, generator=g, device="cuda")
    got.backward(go)
    want.backward(go)
    assert torch.allclose(flow.grad, fr.grad, rtol=1e-5, atol=1e-6)


def test_scatter_accumulators_are_rounded_once_and_left_zero():
    """kernels.cast_clear (vmg_cast_clear) and the accumulator pool (round 4): the fp32 sums of the flow-warp backward and of the attention's gradient banks are
    rounded by one pass that also clears the buffer, so the next scatter needs no fill.  The rounding equals torch's; a pooled buffer is all zero at rest,
    whatever ran before; two backward calls in a row give the same bits as with fresh zero-filled accumulators."""
    from vmg_amd import functional as FH, kernels as K
    g = torch.Generator(device="cuda").manual_seed(31)
    acc = torch.randn((3, 5, 7, 16), generator=g, device="cuda")
    add = torch.randn((3, 5, 7, 16), generator=g, device="cuda").to(torch.bfloat16)
    want = (acc + add.float()).to(torch.bfloat16)
    want0 = acc.to(torch.bfloat16)
    a1, a2 = acc.clone(), acc.clone()
    assert torch.equal(K.cast_clear(a1, torch.bfloat16, add=add), want) and float(a1.abs().max()) == 0.0
    assert torch.equal(K.cast_clear(a2, torch.bfloat16), want0) and float(a2.abs().max()) == 0.0
    K.ACC_POOL.clear()
    x = torch.randn((2, 16, 16, 32), generator=g, device="cuda").to(torch.bfloat16).requires_grad_(True)
    flow = (4.0 * torch.randn((2, 16, 16, 2), generator=g, device="cuda")).requires_grad_(True)
    go = torch.randn((2, 16, 16, 32), generator=g, device="cuda").to(torch.bfloat16)
    grads = []
    for _ in range(3):
        x.grad = flow.grad = None
        FH.grid_sample_flow(x, flow, "bilinear", "border").backward(go)
        grads.append((x.grad.clone(), flow.grad.clone()))
        pooled = K.ACC_POOL.free[((2, 16, 16, 32), str(x.device))]
        assert len(pooled) == 1 and float(pooled[0].abs().max()) == 0.0
    K.ACC_POOL.clear()
    x.grad = flow.grad = None
    FH.grid_sample_flow(x, flow, "bilinear", "border").backward(go)  # a fresh accumulator
    for gx, gf in grads:
        # (float atomics: the arrival order may differ between launches by the last bits of the fp32 sum -- at most one bf16 ulp after the rounding)
        assert float((gx.float() - x.grad.float()).abs().max()) <= 2 ** -7 * float(x.grad.float().abs().max())
        assert torch.allclose(gf, flow.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,t", [(1, 1), (2, 5), (4, 7)])
def test_step_tensors_of_the_lockstep_sweeps_fwd_bwd(dtype, n, t):
    """functional.pair_frame_steps / unpair_steps (vmg_pair_steps, round 4): the recurrence's per-step tensors as separate allocations.
    pair_frame_steps(x)[j] = pair_frames(x)[j] and its gradient = the sum of the two step gradients per frame (fp32 sum, one rounding);
    unpair_steps(steps) = the reference's frame-ordered stacks of the two sweeps (trajectory.py:394-395, 479) and its gradient the matching
    halves -- against the torch spelling (split / reversed / stack), bit-exact."""
    from vmg_amd import functional as FH
    g = torch.Generator(device="cuda").manual_seed(12)
    x = torch.randn((n, t, 6, 5, 16), generator=g, device="cuda").to(dtype).requires_grad_(True)
    steps = FH.pair_frame_steps(x)
    want = FH.pair_frames(x.detach())
    assert len(steps) == t and all(torch.equal(s, w) for s, w in zip(steps, want.unbind(0)))
    gos = [torch.randn(steps[0].shape, generator=g, device="cuda").to(dtype) for _ in range(t)]
    torch.autograd.backward(list(steps), gos)
    go = torch.stack(gos, 0).float()
    ref = (go[:, :n].flip(0) + go[:, n:]).transpose(0, 1)
    assert torch.equal(x.grad, ref.to(dtype))

    feats = [torch.randn((2 * n, 6, 5, 16), generator=g, device="cuda").to(dtype).requires_grad_(True) for _ in range(t)]
    back, fwd = FH.unpair_steps(feats, n)
    fr = [f.detach().clone().requires_grad_(True) for f in feats]
    halves = [f.split(n, 0) for f in fr]
    wback = torch.stack([hv[0] for hv in reversed(halves)], 1)
    wfwd = torch.stack([hv[1] for hv in halves], 1)
    assert torch.equal(back, wback) and torch.equal(fwd, wfwd)
    gb = torch.randn(back.shape, generator=g, device="cuda").to(dtype)
    gf = torch.randn(fwd.shape, generator=g, device="cuda").to(dtype)
    torch.autograd.backward([back, fwd], [gb, gf])
    torch.autograd.backward([wback, wfwd], [gb, gf])
    for a, b in zip(feats, fr):
        assert torch.equal(a.grad, b.grad)
    # only one of the two outputs used: the other half of every step gradient is zero
    feats2 = [f.detach().clone().requires_grad_(True) for f in feats]
    b2, _ = FH.unpair_steps(feats2, n)
    b2.backward(gb)
    for j, f in enumerate(feats2):
        assert torch.equal(f.grad[:n], gb[:, t - 1 - j]) and float(f.grad[n:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(1, 1, 1), (2, 5, 7), (3, 32, 48)])
def test_spynet_level_glue_fwd_bwd(dtype, shape):
    """functional.spy_operand / spy_flow_add (vmg_spy_operand_fwd / _bwd, vmg_spy_flow_add; models/vmg.py:76-85): the level's network input and the
    flow update, against the torch spelling they replace (cast + cat / float + add) -- bit-exact forward and backward."""
    from vmg_amd import functional as FH
    n, h, w = shape
    g = torch.Generator(device="cuda").manual_seed(21)
    ref = torch.randn((n, h, w, 8), generator=g, device="cuda").to(dtype)
    warped = torch.randn((n, h, w, 8), generator=g, device="cuda").to(dtype).requires_grad_(True)
    up = torch.randn((n, h, w, 2), generator=g, device="cuda").requires_grad_(True)
    got = FH.spy_operand(ref, warped, up)
    w2, u2 = warped.detach().clone().requires_grad_(True), up.detach().clone().requires_grad_(True)
    want = torch.cat([ref[..., :3], w2[..., :3], u2.to(dtype)], -1)
    assert torch.equal(got, want)
    go = torch.randn(want.shape, generator=g, device="cuda").to(dtype)
    got.backward(go)
    want.backward(go)
    assert torch.equal(warped.grad, w2.grad) and torch.equal(up.grad, u2.grad)
    res = torch.randn((n, h, w, 2), generator=g, device="cuda").to(dtype).requires_grad_(True)
    up3 = up.detach().clone().requires_grad_(True)
    f = FH.spy_flow_add(up3, res)
    r2, u4 = res.detach().clone().requires_grad_(True), up.detach().clone().requires_grad_(True)
    fw = u4 + r2.float()
    assert torch.equal(f, fw)
    gf = torch.randn(fw.shape, generator=g, device="cuda")
    f.backward(gf)
    fw.backward(gf)
    assert torch.equal(res.grad, r2.grad) and torch.equal(up3.grad, u4.grad)


@pytest.mark.parametrize("shape", [(3, 1, 1, 2), (2, 2, 2, 2), (2, 5, 3, 2), (4, 16, 16, 2), (1, 32, 20, 4)])
def test_flow_upsampling_x2_align_corners_fwd_bwd(shape):
    """SPyNet's flow between pyramid levels (models/vmg.py:97-102): scale * F.interpolate(flow, scale_factor=2, mode='bilinear',
    align_corners=True), and its adjoint.  The backward is a gather (round 4: no zero-fill, no atomics): equal to torch's autograd within fp32
    rounding AND the same bits on every call, whatever the output buffer held before (a 1x1 map and odd sizes included)."""
    import torch.nn.functional as F
    from oracle import recipe as R
    from vmg_amd import kernels as K
    n, h, w, c = shape
    x = R.seeded(shape, 910).cuda()
    g = R.seeded((n, 2 * h, 2 * w, c), 911).cuda()
    xr = x.clone().permute(0, 3, 1, 2).requires_grad_(True)
    want = 2.0 * F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=True)
    want.backward(g.permute(0, 3, 1, 2))
    got = K.upsample2x_ac(x, 2.0)
    assert float((got - want.detach().permute(0, 2, 3, 1)).abs().max()) <= 1e-5
    # poison the allocator's free blocks: the backward must not depend on what its output buffer held
    junk = [torch.full((n * h * w * c,), float("nan"), device="cuda") for _ in range(16)]
    del junk
    dx = K.upsample2x_ac(g, 2.0, backward=True)
    assert torch.isfinite(dx).all()
    assert float((dx - xr.grad.permute(0, 2, 3, 1)).abs().max()) <= 1e-5 * max(1.0, float(xr.grad.abs().max()))
    assert torch.equal(dx, K.upsample2x_ac(g, 2.0, backward=True))
