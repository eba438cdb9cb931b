"""Whole-model parity: vmg_amd.VMG (HIP path, through the C-ABI) vs the CPU oracle and the reference fixtures.

fp32 tolerance: max |diff| <= 2e-3 on [0,1]-scale outputs and identical uint8 images up to PSNR >= 60 dB
(north_star: within 1e-3 PSNR of the reference; both are compared against the same synthetic target below).
bf16 tolerance (stated): PSNR(bf16 HIP, fp32 oracle) >= 40 dB on [0,1] outputs and
|PSNR(hip, target) - PSNR(oracle, target)| <= 0.05 dB.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _run(name, dtype, calls):
    from oracle import cases as C
    from oracle import recipe as R
    from tests.util import build_product
    case = C.CASES[name]
    cfg = case["cfg"]
    shapes, ref_outs = C.load_fixture(os.path.join(GOLD, f"{name}.npz"))
    sd = C.case_state_dict(case, shapes)
    inp = case["inputs"]()
    m = build_product(cfg, dtype)
    missing = m.load_state_dict(sd, strict=True)
    m.eval()
    with torch.no_grad():
        got = [m(inp["x"].cuda()).cpu() for _ in range(calls)]
        want = case["run"]({k: v.clone() for k, v in sd.items()}, inp)[:calls]
    return got, want, ref_outs, inp


@pytest.mark.parametrize("name,calls", [("vmg_tiny_few", 2), ("vmg_tiny_multi", 1), ("vmg_tiny_swin", 1), ("vmg_reds_few_cfg1", 1),
                                        ("vmg_reds_full", 1), ("vmg_tiny_mirror", 1)])
def test_vmg_fp32_matches_oracle_and_reference(name, calls):
    from oracle import cases as C
    from oracle import recipe as R
    from tests.util import psnr
    got, want, ref_outs, inp = _run(name, torch.float32, calls)
    tgt = R.synthetic_target(inp["x"])
    for i, (g, w) in enumerate(zip(got, want)):
        err = float((g - w).abs().max())
        assert err <= 2e-3, f"{name} call {i + 1}: max |hip - oracle| = {err}"
        assert psnr(g, w) >= 60.0
        assert abs(psnr(g, tgt) - psnr(w, tgt)) <= 1e-3
        # and directly against the numbers the unmodified reference produced
        sub = C.subsample(g)
        assert float(np.abs(sub - ref_outs[i]["sub"]).max()) <= 2e-3


def test_vmg_statefulness_call2_differs():
    got, want, _, _ = _run("vmg_tiny_few", torch.float32, 2)
    assert float((got[0] - got[1]).abs().max()) > 0  # T1: weights decayed between the calls
    assert float(((got[1] - got[0]) - (want[1] - want[0])).abs().max()) <= 1e-3


def test_mirrored_clip_takes_the_mirror_branch():
    """Even T with the second half the reversed first half: backward flows are the flipped forward flows (models/vmg.py:426-432,
    :457-462); the product must take that branch (one SPyNet direction) and still match the oracle (case vmg_tiny_mirror above)."""
    from oracle import cases as C
    from tests.util import build_product
    case = C.CASES["vmg_tiny_mirror"]
    m = build_product(case["cfg"], torch.float32)
    m.eval()
    with torch.no_grad():
        m(case["inputs"]()["x"].cuda())
    assert m.frames_mirror is True
    with torch.no_grad():
        m(C.CASES["vmg_tiny_swin"]["inputs"]()["x"].cuda())  # T = 4, not mirrored
    assert m.frames_mirror is False


@pytest.mark.parametrize("name", ["vmg_tiny_few", "vmg_reds_few_cfg1", "vmg_reds_full"])
def test_vmg_bf16_tolerance(name):
    from oracle import recipe as R
    from tests.util import psnr
    got, want, _, inp = _run(name, torch.bfloat16, 1)
    tgt = R.synthetic_target(inp["x"])
    p = psnr(got[0], want[0])
    assert p >= 40.0, f"PSNR(bf16 hip, fp32 oracle) = {p}"
    assert abs(psnr(got[0], tgt) - psnr(want[0], tgt)) <= 0.05


def test_batched_retention_decay_equals_the_per_module_decay():
    """T1: VMG.forward applies the Gamma decay of all MorphFC mixers in one launch (functional.decay_weights_and_repack) and rebuilds their packs
    in one more; three consecutive calls must give the same outputs and leave the same mlp_h / mlp_w weights as the module-by-module path
    (an empty mixer list switches the batching off) -- the weights bit for bit."""
    from oracle import cases as C
    from tests.util import build_product
    case = C.CASES["vmg_tiny_few"]
    shapes, _ = C.load_fixture(os.path.join(GOLD, "vmg_tiny_few.npz"))
    sd = C.case_state_dict(case, shapes)
    x = case["inputs"]()["x"].cuda().to(torch.bfloat16)
    outs, weights = [], []
    for batched in (True, False):
        m = build_product(case["cfg"], torch.bfloat16)
        m.load_state_dict(sd)
        m.eval()
        if not batched:
            m.__dict__["_morph_mixers"] = []
        with torch.no_grad():
            ys = [m(x).float().clone() for _ in range(3)]
        outs.append(ys)
        weights.append({k: v.clone() for k, v in m.state_dict().items() if "mlp_h.0.weight" in k or "mlp_w.0.weight" in k})
    assert weights[0] and weights[0].keys() == weights[1].keys()
    for k in weights[0]:
        assert torch.equal(weights[0][k], weights[1][k]), k  # (the decayed weights themselves: bit for bit)
    for a, b in zip(outs[0], outs[1]):  # (outputs: bit for bit too -- the forward pass has no atomics, the pooled sums are ordered)
        assert torch.equal(a, b)
    assert float((outs[0][0] - outs[0][2]).abs().max()) > 0  # (the decay is stateful: a later call sees smaller weights)
