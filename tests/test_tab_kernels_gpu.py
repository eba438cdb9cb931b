"""HIP kernels of the TAB mixer tail (channel attention, branch re-weighting, tanh gate) vs the oracle, fwd + bwd."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _q(t, dtype):
    return t.to(dtype).float() if dtype == torch.bfloat16 else t


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rcab_matches_oracle_fwd_bwd(dtype):
    """RCAB = 2 convs + channel attention + residual; compares the product module against oracle.rcab (autograd)."""
    from oracle import cases as C, recipe as R, vmg_oracle as O
    from vmg_amd.model import RCAB
    from vmg_amd import functional as FH
    case = C.CASES["rcab_c144"]
    shapes, _ = C.load_fixture("tests/golden/rcab_c144.npz")
    sd = C.case_state_dict(case, shapes)
    x = _q(case["inputs"]()["x"], dtype)
    osd = {k: (_q(v, dtype) if k.endswith("weight") and v.dim() == 4 and v.shape[-1] == 3 else v.clone()).requires_grad_(True) for k, v in sd.items()}
    xo = x.clone().requires_grad_(True)
    want = O.rcab(osd, "", xo)
    go = _q(R.seeded(tuple(x.shape), 91), dtype)
    wg = torch.autograd.grad(want, [xo] + [osd[k] for k in sorted(osd)], go)
    m = RCAB(144).cuda()
    m.load_state_dict(sd)
    xd = x.cuda().to(dtype).requires_grad_(True)
    got = m(xd)
    got.backward(go.cuda().to(dtype))
    FH.flush_deferred_wgrads()
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert float((got.float().cpu() - want).abs().max()) <= tol * max(1.0, float(want.abs().max()))
    assert float((xd.grad.float().cpu() - wg[0]).abs().max()) <= tol * max(1.0, float(wg[0].abs().max()))
    params = dict(m.named_parameters())
    for k, gw in zip(sorted(osd), wg[1:]):
        err = float((params[k].grad.cpu() - gw).abs().max())
        assert err <= (5e-4 if dtype == torch.float32 else 5e-2) * max(1.0, float(gw.abs().max())), f"{k}: {err}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_reweight_mix_and_gate_fwd_bwd(dtype):
    from oracle import recipe as R
    from vmg_amd import functional as FH
    import torch.nn.functional as F
    B, T, H, W, C = 2, 3, 10, 12, 144
    hs = [_q(R.seeded((B, T, H, W, C), 100 + i), dtype).requires_grad_(True) for i in range(3)]
    fc1w, fc1b = R.seeded((36, C), 104, C ** -0.5).requires_grad_(True), R.seeded((36,), 105, 0.1).requires_grad_(True)
    fc2w, fc2b = R.seeded((3 * C, 36), 106, 36 ** -0.5).requires_grad_(True), R.seeded((3 * C,), 107, 0.1).requires_grad_(True)
    x = _q(R.seeded((B, T, H, W, C), 108), dtype).requires_grad_(True)
    # oracle: models/function.py:791-793, 801-802
    a = (hs[0] + hs[1] + hs[2]).mean((1, 2, 3))
    a = F.linear(F.gelu(F.linear(a, fc1w, fc1b)), fc2w, fc2b).reshape(B, C, 3).softmax(-1)[:, None, None, None]
    y = hs[0] * a[..., 0] + hs[1] * a[..., 1] + hs[2] * a[..., 2]
    want = (x + y) * torch.tanh(y)
    go = _q(R.seeded((B, T, H, W, C), 109), dtype)
    leaves = hs + [fc1w, fc1b, fc2w, fc2b, x]
    wg = torch.autograd.grad(want, leaves, go)
    dev = [t.detach().cuda().to(dtype if t.dim() == 5 else torch.float32).requires_grad_(True) for t in leaves]
    yd = FH.reweight_mix(dev[0], dev[1], dev[2], dev[3], dev[4], dev[5], dev[6])
    got = FH.tanh_gate(dev[7], yd)
    gg = torch.autograd.grad(got, dev, go.cuda().to(dtype))
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    assert float((got.float().cpu() - want).abs().max()) <= tol * max(1.0, float(want.abs().max()))
    for i, (g1, g2) in enumerate(zip(gg, wg)):
        assert float((g1.float().cpu() - g2).abs().max()) <= (5e-4 if dtype == torch.float32 else 5e-2) * max(1.0, float(g2.abs().max())), i


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_residual_drop_path(dtype):
    """res + DropPath(y) * s in one kernel: timm semantics (per-sample mask of 0 or 1/keep), eval / p = 0 is a plain add, and the
    gradients are dy (residual) and dy * mask * s (branch)."""
    from vmg_amd import functional as FH
    torch.manual_seed(0)
    B, T, H, W, C = 6, 2, 8, 8, 144
    res = torch.randn(B, T, H, W, C, device="cuda").to(dtype).requires_grad_(True)
    y = torch.randn(B, T, H, W, C, device="cuda").to(dtype).requires_grad_(True)
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    out = FH.residual_drop_path(res, y, 0.0, True, 1.0)
    assert float((out.float() - (res.float() + y.float())).abs().max()) <= tol
    out = FH.residual_drop_path(res, y, 0.3, False, 0.5)
    assert float((out.float() - (res.float() + 0.5 * y.float())).abs().max()) <= tol
    p, s = 0.4, 0.5
    out = FH.residual_drop_path(res, y, p, True, s)
    ratio = ((out.float() - res.float()) / y.float()).detach()
    seen = set()
    for b in range(B):
        r = float(ratio[b].median())
        assert abs(r) <= tol or abs(r - s / (1 - p)) <= 5 * tol + 1e-3, r
        assert float((ratio[b] - r).abs().median()) <= 5 * tol + 1e-3
        seen.add(round(r, 2))
    dy = torch.randn_like(out)
    out.backward(dy)
    assert torch.equal(res.grad, dy)
    m = ratio.reshape(B, -1).median(1).values.reshape(B, 1, 1, 1, 1)
    assert float((y.grad.float() - dy.float() * m).abs().max()) <= 5 * tol * max(1.0, float(dy.abs().max())) + 1e-3


@pytest.mark.parametrize("cfg", [(144, 8, 24, 20), (144, 16, 20, 36), (112, 8, 16, 16), (16, 8, 12, 10), (32, 16, 40, 24), (64, 8, 8, 8)])
@pytest.mark.parametrize("axis", ["h", "w"])
def test_fused_morphfc_branch_fwd_bwd(cfg, axis):
    """vmg_morphfc_fwd (token reshuffle inside the GEMM addressing) vs the oracle's restatement of the reference lines (pad, reshuffle,
    Linear + ReLU, / Cp, inverse reshuffle, crop; models/function.py:763-786): output, input gradient, weight / bias gradients.  Sizes
    that are no multiple of the chunk exercise the padded positions; bf16 tolerance as for the convs (inputs / weights pre-rounded)."""
    import math
    import torch.nn.functional as F
    from oracle import recipe as R, vmg_oracle as O
    from vmg_amd import functional as FH, kernels as K
    C, chunk, H, W = cfg
    Cp = int(math.ceil(C / chunk)) * chunk
    dt = torch.bfloat16
    x = _q(R.seeded((2, 3, H, W, C), 110), dt)
    w = _q(R.seeded((Cp, Cp), 111, Cp ** -0.5), dt)
    b = R.seeded((Cp,), 112, 0.1)
    go = _q(R.seeded((2, 3, H, W, C), 113), dt)
    xo, wo, bo = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    t = O.morph_tokens(xo, axis, chunk, Cp)
    want = O.morph_untokens(F.relu(F.linear(t, wo, bo)) / Cp, axis, chunk, Cp, H, W, C)
    wg = torch.autograd.grad(want, [xo, wo, bo], go)
    assert K.morph_fused_ok(x.cuda().to(dt), chunk, Cp)
    xd = x.cuda().to(dt).requires_grad_(True)
    wd = torch.nn.Parameter(w.cuda())
    bd = torch.nn.Parameter(b.cuda())
    got = FH.morph_linear(xd, wd, bd, axis, chunk, Cp)
    got.backward(go.cuda().to(dt))
    sc = max(1e-3, float(want.abs().max()))
    assert float((got.float().cpu() - want).abs().max()) <= 2e-2 * sc
    assert float((xd.grad.float().cpu() - wg[0]).abs().max()) <= 3e-2 * max(1e-6, float(wg[0].abs().max()))
    assert float((wd.grad.cpu() - wg[1]).abs().max()) <= 3e-2 * max(1e-6, float(wg[1].abs().max()))
    assert float((bd.grad.cpu() - wg[2]).abs().max()) <= 3e-2 * max(1e-6, float(wg[2].abs().max()))


@pytest.mark.parametrize("cfg", [(144, 8, 24, 20), (112, 8, 13, 16), (32, 16, 40, 24)])
@pytest.mark.parametrize("axis", ["h", "w"])
def test_fused_morphfc_token_side_outputs(cfg, axis):
    """The token matrices the fused kernel writes next to its output are exactly the reference's reshuffle (FH.morph_tokens, checked against the
    oracle elsewhere) of x in the forward call and of dy * relu'(y) / Cp in the data-gradient call; rows past the last group are zero."""
    import math
    from oracle import recipe as R
    from vmg_amd import functional as FH, kernels as K, hip
    C, chunk, H, W = cfg
    Cp = int(math.ceil(C / chunk)) * chunk
    dt = torch.bfloat16
    x = R.seeded((1, 3, H, W, C), 210).cuda().to(dt)
    dy = R.seeded((1, 3, H, W, C), 211).cuda().to(dt)
    w = R.seeded((Cp, Cp), 212, Cp ** -0.5).cuda()
    nct = (Cp + 15) // 16
    y, tok = K.morphfc_forward(x, axis, chunk, Cp, FH.packed(w, dt, "fwd", [Cp], tiles=nct), None, True, 1.0, 1.0 / Cp, want_tokens=True)
    ref = FH.morph_tokens(x, axis, chunk, Cp).reshape(-1, Cp)
    assert tok.shape[0] >= ref.shape[0] and tok.shape[0] % 16 == 0
    assert torch.equal(tok[:ref.shape[0]], ref) and not bool(tok[ref.shape[0]:].any())
    _, dtok = K.morphfc_forward(dy, axis, chunk, Cp, FH.packed(w, dt, "dgrad", None, 0, Cp, tiles=nct), None, False, 1.0 / Cp, 1.0, mask=y, want_tokens=True)
    dref = FH.morph_tokens(K.act_backward(dy, y, hip.ACT_RELU, 0.0, 1.0 / Cp), axis, chunk, Cp).reshape(-1, Cp)
    assert torch.equal(dtok[:dref.shape[0]], dref) and not bool(dtok[dref.shape[0]:].any())


@pytest.mark.parametrize("case", [(28, 144, 18, 144, "relu", 0), (4, 144, 36, 432, "gelu", 1), (7, 448, 56, 448, "relu", 0), (1, 112, 28, 336, "gelu", 1),
                                  (70, 32, 4, 96, "gelu", 1)])
def test_se_mlp_fwd_bwd_matches_torch(case):
    """The one-launch squeeze-excite MLPs (CALayer.conv_du: ReLU + sigmoid; MorphFC reweight: GELU + softmax over the three branches)
    against torch autograd on the same fp32 values: outputs, input gradient (with the 1/R scale) and all four parameter gradients."""
    import torch.nn.functional as F
    from oracle import recipe as R
    from vmg_amd import hip, kernels as K
    G, C, Hd, Co, act, mode = case
    m = R.seeded((G, C), 11).requires_grad_(True)
    w1 = R.seeded((Hd, C), 12, C ** -0.5).requires_grad_(True)
    b1 = R.seeded((Hd,), 13, 0.1).requires_grad_(True)
    w2 = R.seeded((Co, Hd), 14, Hd ** -0.5).requires_grad_(True)
    b2 = R.seeded((Co,), 15, 0.1).requires_grad_(True)
    dout = R.seeded((G, Co), 16)
    z1 = F.relu(F.linear(m, w1, b1)) if act == "relu" else F.gelu(F.linear(m, w1, b1))
    z2 = F.linear(z1, w2, b2)
    want = torch.sigmoid(z2) if mode == 0 else z2.reshape(G, Co // 3, 3).softmax(-1).reshape(G, Co)
    grads = torch.autograd.grad(want, (m, w1, b1, w2, b2), dout)
    code = hip.ACT_RELU if act == "relu" else hip.ACT_GELU
    dev = lambda t: t.detach().cuda()
    pre, out = K.se_mlp_forward(dev(m), dev(w1), dev(b1), dev(w2), dev(b2), code, mode)
    assert float((out.cpu() - want.detach()).abs().max()) <= 2e-6
    got = K.se_mlp_backward(dout.cuda(), out, dev(m), pre, dev(w1), dev(w2), code, mode, 0.25)
    for g, w, name, sc in zip(got, grads, ("dm", "dw1", "db1", "dw2", "db2"), (0.25, 1, 1, 1, 1)):
        err = float((g.cpu() - sc * w).abs().max())
        assert err <= 1e-5 * max(1.0, float(w.abs().max())), f"{name}: {err}"


def test_drop_path_plan_matches_per_call_semantics():
    """functional._DropPlan: the second forward pass draws all DropPath masks at once.  Every (sample, call) coefficient must be 0 or
    scale / keep (timm's DropPath), constant over channels and pixels; a call that deviates from the recorded sequence falls back."""
    from vmg_amd import functional as FH
    B, T, H, W, C = 6, 2, 4, 4, 16
    res = torch.zeros(B, T, H, W, C, device="cuda", dtype=torch.bfloat16)
    y = torch.ones(B, T, H, W, C, device="cuda", dtype=torch.bfloat16)
    calls = [(0.25, 1.0), (0.5, 0.5), (0.1, 1.0)]
    for rnd in range(3):
        FH.DROP.begin(res.device, True)
        for p, s in calls:
            out = FH.residual_drop_path(res, y, p, True, s).float()
            per = out.reshape(B, -1)
            assert bool((per == per[:, :1]).all())  # one coefficient per sample
            v = per[:, 0]
            want = s / (1.0 - p)
            assert bool(((v == 0) | ((v - want).abs() <= 1e-2 * want)).all()), (rnd, p, s, v)
        if rnd >= 1:
            assert FH.DROP.plan is not None and FH.DROP.idx == len(calls)
    FH.DROP.begin(res.device, True)
    FH.residual_drop_path(res, y, 0.3, True, 1.0)  # not the recorded first call: per-call path, still valid
    assert FH.DROP.g is None
    FH.DROP.begin(res.device, False)


def test_drop_path_plan_with_calls_of_different_shapes():
    """The full configuration's stages differ in channel count (112 / 224 / 448) and its residual sites alternate between residual_drop_path
    and gate_residual: the plan must cover such a sequence too (round 4: one flat coefficient buffer), not fall back to four tiny kernels per call."""
    from vmg_amd import functional as FH
    shapes = [(3, 16), (3, 32), (5, 32), (3, 16)]
    calls = [(0.25, 1.0), (0.5, 0.5), (0.1, 1.0), (0.4, 2.0)]
    ts = [(torch.zeros(B, 2, 4, 4, C, device="cuda", dtype=torch.bfloat16), torch.ones(B, 2, 4, 4, C, device="cuda", dtype=torch.bfloat16)) for B, C in shapes]
    zeros_seen = 0
    for rnd in range(4):
        FH.DROP.begin(ts[0][0].device, True)
        for i, ((p, s), (res, y)) in enumerate(zip(calls, ts)):
            if i % 2:
                out = FH.gate_residual(y, torch.zeros_like(y) + 0.5, res, p, True, s).float() / (1.5 * float(torch.tanh(torch.tensor(0.5))))
            else:
                out = FH.residual_drop_path(res, y, p, True, s).float()
            per = out.reshape(out.shape[0], -1)
            assert bool((per == per[:, :1]).all())
            v = per[:, 0]
            want = s / (1.0 - p)
            assert bool(((v == 0) | ((v - want).abs() <= 2e-2 * want)).all()), (rnd, i, v, want)
            zeros_seen += int((v == 0).sum())
        if rnd >= 1:
            assert FH.DROP.plan is not None and FH.DROP.g is not None and FH.DROP.idx == len(calls)
    assert zeros_seen > 0  # (56 draws with drop probabilities 0.1 .. 0.5)
    FH.DROP.begin(ts[0][0].device, False)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("axis", ["h", "w"])
@pytest.mark.parametrize("geom", [(1, 2, 16, 16, 224, 12, 228), (1, 3, 32, 32, 224, 16, 224), (2, 1, 8, 8, 448, 8, 448), (1, 2, 10, 7, 40, 4, 44), (1, 1, 5, 9, 16, 3, 18)])
def test_morph_token_gather_scatter_general_path(dtype, axis, geom):
    """The general MorphFC path's two kernels (round 4; the full configuration's chunk 16 / 12 / 8 with Cp 224 / 228 / 448, models/function.py:
    749-750, 763-764, 772, 776-777, 785) against the torch spelling of the reference's pad + rearrange chain (functional.morph_tokens /
    morph_untokens): bit-exact both ways, zero features up to the padded row length, scatter(gather(x)) == x, and <gather(x), t> == <x, scatter(t)>
    (they are each other's backward)."""
    from oracle import recipe as R
    from vmg_amd import functional as FH, kernels as K
    B, T, H, W, C, chunk, Cp = geom
    x = R.seeded((B, T, H, W, C), 1200).to(dtype).cuda()
    want = FH.morph_tokens(x, axis, chunk, Cp).reshape(-1, Cp)
    ld = (Cp + 7) // 8 * 8
    got = K.morph_tokens_gather(x, axis, chunk, Cp, ld)
    assert got.shape == (want.shape[0], ld)
    assert torch.equal(got[:, :Cp], want)
    assert ld == Cp or float(got[:, Cp:].abs().max()) == 0.0
    t = R.seeded(tuple(want.shape), 1201).to(dtype).cuda()
    back = K.morph_tokens_scatter(t, axis, chunk, Cp, (B, T, H, W, C))
    G = want.shape[0] // chunk // (B * T)
    assert torch.equal(back, FH.morph_untokens(t.reshape(B, T, G, chunk, Cp), axis, chunk, Cp, H, W, C))
    assert torch.equal(K.morph_tokens_scatter(got, axis, chunk, Cp, (B, T, H, W, C)), x)      # row stride ld, the zero features ignored
    a = float((got[:, :Cp].double() * t.double()).sum())
    b = float((x.double() * back.double()).sum())
    assert abs(a - b) <= 1e-9 * max(1.0, abs(a))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("geom", [(1, 2, 16, 16, 224, 12, 228), (1, 2, 32, 32, 224, 16, 224), (1, 2, 8, 8, 448, 8, 448)])
def test_morph_linear_general_path_equals_the_torch_spelling(dtype, geom):
    """functional.morph_linear on the shapes the fused kernel does not cover: output, input gradient and parameter gradients equal the same branch
    spelled with torch copies around the same Linear kernel (what round 3 ran), bit for bit in the forward (the Linear sees the same token matrix)
    and within rounding of the reductions in the backward."""
    from oracle import recipe as R
    from vmg_amd import functional as FH
    from vmg_amd import hip
    B, T, H, W, C, chunk, Cp = geom
    res = []
    for general in (True, False):
        x = R.seeded((B, T, H, W, C), 1210).to(dtype).cuda().requires_grad_(True)
        w = torch.nn.Parameter(R.seeded((Cp, Cp), 1211, Cp ** -0.5).cuda())
        b = torch.nn.Parameter(R.seeded((Cp,), 1212, 0.1).cuda())
        outs = []
        for axis in ("h", "w"):
            if general:
                assert not FH.K.morph_fused_ok(x, chunk, Cp)
                y = FH.morph_linear(x, w, b, axis, chunk, Cp)
            else:
                tk = FH.morph_tokens(x, axis, chunk, Cp)
                y = FH.morph_untokens(FH.linear(tk, w, b, act=hip.ACT_RELU, alpha=1.0 / Cp), axis, chunk, Cp, H, W, C)
            outs.append(y)
        g = R.seeded((B, T, H, W, C), 1213).to(dtype).cuda()
        (outs[0] * g).sum().add((outs[1] * g.flip(2)).sum()).backward()
        res.append(([o.detach() for o in outs], x.grad, w.grad, b.grad))
    for a, bb in zip(res[0][0], res[1][0]):
        assert torch.equal(a, bb)
    for i in (1, 2, 3):
        a, bb = res[0][i].float(), res[1][i].float()
        assert float((a - bb).abs().max()) <= 1e-5 * max(1.0, float(bb.abs().max())) if dtype == torch.float32 else float((a - bb).abs().max()) <= 2e-2 * float(bb.abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("drop", [0.0, 0.3])
def test_gate_residual_one_pass_equals_gate_then_residual(dtype, drop):
    """functional.gate_residual (round 4): res + DropPath((x + y) tanh(y)) * s in one pass each way -- the same bits as tanh_gate followed by
    residual_drop_path (models/function.py:801-802, 1212-1214), forward and all three gradients, with and without dropped samples."""
    from oracle import recipe as R
    from vmg_amd import functional as FH
    shape = (3, 2, 8, 8, 144)
    res = []
    for fused in (False, True):
        x = R.seeded(shape, 1400).to(dtype).cuda().requires_grad_(True)
        y = R.seeded(shape, 1401).to(dtype).cuda().requires_grad_(True)
        r = R.seeded(shape, 1402).to(dtype).cuda().requires_grad_(True)
        torch.manual_seed(5)  # (the per-call DropPath mask: same draw in both runs)
        FH.DROP.begin(x.device, False)
        if fused:
            out = FH.gate_residual(x, y, r, drop, True, 0.5)
        else:
            out = FH.residual_drop_path(r, FH.tanh_gate(x, y), drop, True, 0.5)
        g = R.seeded(shape, 1403).to(dtype).cuda()
        out.backward(g)
        res.append((out.detach(), x.grad, y.grad, r.grad))
    for a, b in zip(*res):
        assert torch.equal(a, b)
