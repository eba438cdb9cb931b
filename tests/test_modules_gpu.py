"""Per-module replays of the reference fixtures through the PRODUCT modules (HIP path), fp32 strict and bf16 with the stated
tolerance: the widths of the full VMG-REDS configuration (112 / 224 channels, chunk 12 with its padded Ch = 228 and Hp = 24,
groups = 4) as well as the 144-channel few_levels ones.  Each case is checked against the oracle on the same weights and
against the numbers the unmodified reference produced (tests/golden/*.npz).

bf16 tolerance (stated): max |diff| <= 3e-2 of the output scale (inputs and weights are rounded to bf16 by the product; the
oracle runs fp32 on the unrounded values)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _build(name):
    from vmg_amd import model as M
    if name.startswith("morphfc"):
        C, ch = (144, 8) if name == "morphfc_c144_chunk8" else (224, 12)
        return M.Enhanced_MorphFCs_decay(C, ch, ch, qkv_bias=True, channel_mixer="rcab")
    if name == "rcab_c144":
        return M.RCAB(144)
    if name == "mlp_cnn_c144":
        return M.Mlp_cnn(144, exp_r=2, n_groups=1)
    if name == "mlp_cnn_c112_g4":
        return M.Mlp_cnn(112, exp_r=2, n_groups=4)
    if name == "tab_c144":
        return M.TAB(144, 8, 8, 2, 1, True, 0.0, "ffn_cnn", 1.0, "rcab")
    if name == "updown_down":
        return M.UpdownkeepSampling(144, 144, "down")
    if name == "updown_up":
        return M.UpdownkeepSampling(144, 144, "up")
    raise KeyError(name)


CALLS = {"morphfc_c144_chunk8": 3, "tab_c144": 2}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["morphfc_c144_chunk8", "morphfc_c224_chunk12", "rcab_c144", "mlp_cnn_c144", "mlp_cnn_c112_g4", "tab_c144",
                                  "updown_down", "updown_up"])
def test_module_matches_oracle_and_reference_fixture(name, dtype):
    from oracle import cases as C
    case = C.CASES[name]
    shapes, ref_outs = C.load_fixture(os.path.join(GOLD, f"{name}.npz"))
    sd = C.case_state_dict(case, shapes)
    x = case["inputs"]()["x"]
    m = _build(name).cuda()
    m.load_state_dict(sd, strict=True)
    m.eval()
    calls = CALLS.get(name, 1)
    with torch.no_grad():
        got = [m(x.cuda().to(dtype)).float().cpu() for _ in range(calls)]  # calls #1.. (the mixer weights decay in place, T1)
        want = case["run"]({k: v.clone() for k, v in sd.items()}, {"x": x})
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    for i, (g, w) in enumerate(zip(got, want)):
        g = g.reshape(w.shape)
        scale = max(1.0, float(w.abs().max()))
        err = float((g - w).abs().max())
        assert err <= tol * scale, f"{name} call {i + 1} ({dtype}): max |hip - oracle| = {err} (scale {scale})"
        sub = C.subsample(g)
        assert float(np.abs(sub - ref_outs[i]["sub"]).max()) <= tol * scale, f"{name} call {i + 1}: differs from the reference fixture"


def test_spynet_matches_reference_fixture():
    """SPyNet through the product module vs the oracle and the reference's own flows (fp32)."""
    import vmg_amd
    from oracle import cases as C
    for name in ("spynet", "spynet_48x40"):
        case = C.CASES[name]
        shapes, ref_outs = C.load_fixture(os.path.join(GOLD, f"{name}.npz"))
        sd = C.case_state_dict(case, shapes)
        inp = case["inputs"]()
        m = vmg_amd.SPyNet(None).cuda()
        m.load_state_dict({k[len("spynet."):]: v for k, v in sd.items()}, strict=True)
        with torch.no_grad():
            got = m(inp["ref"].cuda(), inp["supp"].cuda()).float().cpu()
            want = case["run"]({k: v.clone() for k, v in sd.items()}, inp)[0]
        scale = max(1.0, float(want.abs().max()))
        assert float((got - want).abs().max()) <= 2e-3 * scale
        assert float(np.abs(C.subsample(got) - ref_outs[0]["sub"]).max()) <= 2e-3 * scale


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["updown_down", "updown_up"])
def test_updown_backward_matches_oracle_autograd(name, dtype):
    """UpdownkeepSampling backward (the fused space<->depth + LayerNorm scatters its input gradient): d/dx and every parameter
    gradient vs torch autograd through the oracle."""
    from oracle import cases as C, recipe as R, vmg_oracle as O
    case = C.CASES[name]
    shapes, _ = C.load_fixture(os.path.join(GOLD, f"{name}.npz"))
    sd = C.case_state_dict(case, shapes)
    x = case["inputs"]()["x"]
    if dtype == torch.bfloat16:
        x = x.to(dtype).float()
    mode = "down" if name.endswith("down") else "up"
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xo = x.clone().requires_grad_(True)
    want = O.updown(osd, "", xo, mode)
    go = R.seeded(tuple(want.shape), 95)
    if dtype == torch.bfloat16:
        go = go.to(dtype).float()
    wg = torch.autograd.grad(want, [xo] + [osd[k] for k in sorted(osd)], go)
    m = _build(name).cuda()
    m.load_state_dict(sd)
    xd = x.cuda().to(dtype).requires_grad_(True)
    got = m(xd)
    got.backward(go.cuda().to(dtype).reshape(got.shape))
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert float((got.float().cpu().reshape(want.shape) - want).abs().max()) <= tol * max(1.0, float(want.abs().max()))
    assert float((xd.grad.float().cpu() - wg[0]).abs().max()) <= tol * max(1.0, float(wg[0].abs().max()))
    params = dict(m.named_parameters())
    for k, gw in zip(sorted(osd), wg[1:]):
        err = float((params[k].grad.cpu() - gw).abs().max())
        assert err <= (5e-4 if dtype == torch.float32 else 5e-2) * max(1.0, float(gw.abs().max())), f"{k}: {err}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mdsc_skip_fwd_bwd(dtype):
    """The multi-scale skip (adaptive max-pool /4, conv1x1, GroupNorm(1), ReLU; models/vmg.py:388-400) on the HIP kernels vs the
    oracle's autograd: output, input gradient and all four parameter gradients."""
    import torch.nn as nn
    from oracle import recipe as R, vmg_oracle as O
    from vmg_amd.model import VMG
    B, T, H, W, C, C2 = 2, 2, 16, 24, 32, 64
    x = R.seeded((B, T, H, W, C), 71)
    sd = {"0.weight": R.seeded((C2, C, 1, 1), 72, C ** -0.5), "0.bias": R.seeded((C2,), 73, 0.1), "1.weight": 1 + R.seeded((C2,), 74, 0.1),
          "1.bias": R.seeded((C2,), 75, 0.1)}
    if dtype == torch.bfloat16:  # the product rounds activations and the conv weight to bf16: give the oracle the same values
        x = x.to(dtype).float()
        sd["0.weight"] = sd["0.weight"].to(dtype).float()
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xo = x.clone().requires_grad_(True)
    want = O.mdsc_skip(osd, "", xo)
    go = R.seeded(tuple(want.shape), 76)
    if dtype == torch.bfloat16:
        go = go.to(dtype).float()
    wg = torch.autograd.grad(want, [xo] + [osd[k] for k in sorted(osd)], go)
    seq = nn.Sequential(nn.Conv2d(C, C2, 1, 1, 0), nn.GroupNorm(1, C2), nn.ReLU()).cuda()
    seq.load_state_dict(sd)
    xd = x.cuda().to(dtype).requires_grad_(True)
    got = VMG._mdsc(None, seq, xd)
    got.backward(go.cuda().to(dtype))
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert float((got.float().cpu() - want).abs().max()) <= tol * max(1.0, float(want.abs().max()))
    assert float((xd.grad.float().cpu() - wg[0]).abs().max()) <= tol * max(1.0, float(wg[0].abs().max()))
    params = dict(seq.named_parameters())
    for k, gw in zip(sorted(osd), wg[1:]):
        err = float((params[k].grad.cpu().reshape(gw.shape) - gw).abs().max())
        assert err <= (5e-4 if dtype == torch.float32 else 5e-2) * max(1.0, float(gw.abs().max())), f"{k}: {err}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["swin_w2_t5", "swin_w4_t7"])
def test_swin_decoder_layer_fwd_bwd(name, dtype):
    """swin_3d.DecoderLayer (two blocks, the second shifted) through the product: the window attention kernel folds partition, the
    zero padding (20 x 20 -> 24 x 24 in swin_w2_t5), the roll, the -100 mask and the bias gather into its addressing; T = 5 / 7 exercise
    the frame repetition.  Forward vs the oracle and the reference fixture; input and parameter gradients vs the oracle's autograd."""
    from oracle import cases as C, recipe as R, vmg_oracle as O
    from vmg_amd.model import DecoderLayer
    case = C.CASES[name]
    shapes, ref_outs = C.load_fixture(os.path.join(GOLD, f"{name}.npz"))
    sd = C.case_state_dict(case, shapes)
    ws = case["window_of"]("")
    heads = 4 if name == "swin_w2_t5" else 8
    x = case["inputs"]()["x"]
    if dtype == torch.bfloat16:
        x = x.to(dtype).float()
    osd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    xo = x.clone().requires_grad_(True)
    want = O.swin_decoder_layer(osd, "", xo, heads, ws)
    go = R.seeded(tuple(want.shape), 96)
    if dtype == torch.bfloat16:
        go = go.to(dtype).float()
    leaves = [k for k in sorted(osd) if osd[k].dtype.is_floating_point]
    wg = torch.autograd.grad(want, [xo] + [osd[k] for k in leaves], go)
    m = DecoderLayer(32, 2, heads, list(ws), 2, True).cuda()
    m.load_state_dict(sd, strict=True)
    xd = x.cuda().to(dtype).requires_grad_(True)
    got = m(xd)
    got.backward(go.cuda().to(dtype))
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    scale = max(1.0, float(want.abs().max()))
    assert float((got.float().cpu() - want).abs().max()) <= tol * scale
    if dtype == torch.float32:
        assert float(np.abs(C.subsample(got.float().cpu()) - ref_outs[0]["sub"]).max()) <= tol * scale
    assert float((xd.grad.float().cpu() - wg[0]).abs().max()) <= tol * max(1.0, float(wg[0].abs().max()))
    params = dict(m.named_parameters())
    for k, gw in zip(leaves, wg[1:]):
        err = float((params[k].grad.cpu() - gw).abs().max())
        assert err <= (5e-4 if dtype == torch.float32 else 5e-2) * max(1.0, float(gw.abs().max())), f"{k}: {err}"
