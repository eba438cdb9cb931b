"""3-D shifted-window attention kernels (models/swin_3d.py:167-252): the MFMA form (round 4: QK^T, PV and their gradients on
v_mfma_f32_16x16x32_bf16 for bf16 tensors) against the VALU form (one thread per token, fp32 arithmetic on the same bf16 inputs), which the
module-level tests pin to the oracle and the reference fixtures (tests/test_modules_gpu.py::test_swin_decoder_layer_fwd_bwd)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("geom", [(1, 4, 16, 16, 144, 8, 4, (0, 0, 0)), (1, 4, 16, 16, 144, 8, 4, (2, 4, 4)), (2, 5, 8, 16, 32, 4, 2, (1, 4, 4)),
                                  (1, 2, 20, 20, 32, 4, 2, (0, 0, 0)), (1, 2, 20, 12, 32, 4, 2, (1, 4, 4)), (1, 8, 8, 8, 64, 8, 4, (2, 0, 0)),
                                  (1, 6, 8, 8, 48, 2, 6, (3, 4, 4)), (1, 8, 8, 16, 64, 4, 8, (4, 4, 4))])
def test_win3d_mfma_matches_the_valu_kernel(geom):
    """Forward output and log-sum-exp, and every gradient (q, kv, bias table, the q / kv Linear biases through padded positions), of the MFMA
    kernels vs the VALU kernel on the same bf16 tensors.  Shapes: head dimensions 18, 8, 4, 24, 16; temporal windows 2, 4, 6, 8; shifted and
    unshifted blocks; frames and maps that need padding (D = 5 with wt = 2, 20 x 20, 20 x 12).  Stated: output within 2e-2 of its scale (the
    probabilities enter the PV product as bf16), lse within 1e-3 of its scale (a padded position's q / k is the Linear's bias: the VALU kernel takes
    it in fp32, the MFMA operands round it to bf16 like every real token's), gradients within 3e-2 of each tensor's scale."""
    from oracle import recipe as R
    from vmg_amd import hip, kernels as K
    B, D, H, W, C, heads, wt, shift = geom
    dt = torch.bfloat16
    q = R.seeded((B, D, H, W, C), 1300, 0.7).to(dt).cuda()
    kv = R.seeded((B, D, H, W, 2 * C), 1301, 0.7).to(dt).cuda()
    bq = R.seeded((C,), 1302, 0.3).cuda()
    bkv = R.seeded((2 * C,), 1303, 0.3).cuda()
    table = R.seeded(((2 * wt - 1) * 225, heads), 1304, 0.5).cuda()
    dout = R.seeded((B, D, H, W, C), 1305).to(dt).cuda()
    lib = hip.lib()
    prev = lib.vmg_win3d_variant(-1)
    res = []
    try:
        for variant in (0, 1):
            lib.vmg_win3d_variant(variant)
            out, lse = K.win3d_attn_forward(q, kv, bq, bkv, table, heads, wt, shift)
            dq, dkv, dtable, dbq, dbkv = K.win3d_attn_backward(q, kv, bq, bkv, table, out, lse, dout, heads, wt, shift)
            res.append([t.float().cpu() for t in (out, lse, dq, dkv, dtable, dbq, dbkv)])
    finally:
        lib.vmg_win3d_variant(prev)
    assert prev == 1
    names = ["out", "lse", "dq", "dkv", "dtable", "dbq", "dbkv"]
    tols = [2e-2, 1e-3, 3e-2, 3e-2, 3e-2, 3e-2, 3e-2]
    for n, tol, a, b in zip(names, tols, res[0], res[1]):
        assert torch.isfinite(b).all(), n
        scale = max(float(a.abs().max()), 1e-6)
        err = float((a - b).abs().max())
        assert err <= tol * scale, f"{n}: max |valu - mfma| = {err:.3e} at scale {scale:.3e} ({geom})"
