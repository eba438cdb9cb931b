"""Parity of the BENCHMARKED configuration itself: bench.py runs VMG-REDS-few_levels on 4 clips x 7 frames x 64x64
in bf16 (BASELINE configs[1]).  At that batch the recurrent residual chains see M = 2 * 4 * 64 * 64 = 32 768 pixels per launch and take the
weight-streaming route (functional.choose_tiling -> deep = 3: vmg_resblock_chain_fwd / _bwd over conv_ws_kernel<9>, the kernel the bench's
roofline object prices), where every B = 1 test takes the K-split route.  These cases put that route -- the chain descriptor array over the
weight-streaming kernel, pair_frames at n = 4 inside the model, the inference tile size -- under the oracle.

Reference: models/trajectory.py:16-52 (ResidualBlocksWithInputConv), :165-221 (ResidualBlockNoBN0), :300-490 (Trajectory_multi_head.forward),
models/vmg.py:585-637 (VMG.forward)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _chain_counts():
    from vmg_amd import kernels as K
    return {d: dict(v) for d, v in K.CHAIN_STATS.items()}


def _delta(before, after, direction, deep):
    return after[direction].get(deep, 0) - before[direction].get(deep, 0)


def test_bench_batch_bf16_forward_loss_and_gradients_vs_oracle():
    """The bench's own job: few_levels, B = 4, T = 7, 64x64, bf16, train mode (DropPath rates 0 so that the pass is deterministic), the bench's
    own loss (Charbonnier + 0.005 x Laplacian-edge, utils/loss.py:22-79) on the HIP loss kernels, deferred batched weight gradients -- against the
    fp32 oracle's forward, loss and autograd on the same weights and clip.  Stated bf16 tolerance: PSNR(hip, oracle) >= 40 dB,
    loss within 2 %, cosine of the concatenated gradient >= 0.999, per parameter tensor (norm >= 1e-3 of the largest) relative L2 <= 0.04 for
    the network proper and the SPyNet bound of tests/test_grad_gpu.py.  Asserts that all 28 chain calls (2 modules x 7 steps, both sweeps in
    lockstep, forward and backward) ran the weight-streaming route."""
    from oracle import cases as C
    from oracle import recipe as R
    from oracle import vmg_oracle as O
    from tests.util import build_product, psnr
    from tests.test_grad_gpu import SPYNET_BF16_BOUND
    from vmg_amd import functional as FH
    from vmg_amd.train import charbonnier_edge_loss_hip
    B, T = 4, 7
    cfg = C.cfg_reds_few(T=T)
    assert FH.choose_tiling(2 * B * 64 * 64, 144, 3, torch.bfloat16, [144])[2] == 3   # the weight-streaming kernel
    assert FH.choose_tiling(2 * 1 * 64 * 64, 144, 3, torch.bfloat16, [144])[2] == 2   # (what every B = 1 case takes)
    shapes, _ = C.load_fixture(os.path.join(GOLD, "vmg_reds_few_cfg1.npz"))
    chunk_of, window_of = R.vmg_chunk_lookup(cfg)
    sd = R.recipe_state_dict(shapes, 0, chunk_of, window_of)
    x = R.synthetic_clip(B, T, 64, 64, 1234)     # bench.py's clip (rank 0)
    tgt = R.synthetic_target(x)
    before = _chain_counts()
    FH.set_wgrad_mode("deferred")
    try:
        m = build_product(cfg, torch.bfloat16)
        m.load_state_dict(sd)
        m.train()
        out = m(x.cuda())
        loss = charbonnier_edge_loss_hip(out.float(), tgt.cuda(), 1e-12, 0.005)
        loss.backward()
    finally:
        FH.set_wgrad_mode("autograd")
    after = _chain_counts()
    assert _delta(before, after, "fwd", 3) == 2 * T and _delta(before, after, "bwd", 3) == 2 * T, (before, after)
    assert sum(after["fwd"].values()) - sum(before["fwd"].values()) == 2 * T  # ... and no chain call took another route

    osd = {}
    for k, v in sd.items():
        v = v.clone()
        if k.endswith("mlp_h.0.weight"):
            v = v * sd[k.replace("mlp_h.0.weight", "gamma_h")]
        if k.endswith("mlp_w.0.weight"):
            v = v * sd[k.replace("mlp_w.0.weight", "gamma_w")]
        if v.dtype.is_floating_point and not R.is_buffer(k):
            v.requires_grad_(True)
        osd[k] = v
    oout = O.vmg_forward(osd, cfg, x, mutate=False, call_index=0)
    oloss = O.charbonnier_edge_loss(oout, tgt)
    oloss.backward()
    got = out.detach().float().cpu()
    p = psnr(got, oout.detach())
    print(f"bench batch: PSNR(hip bf16, oracle fp32) = {p:.2f} dB, loss {float(loss):.6f} vs {float(oloss):.6f}")
    assert p >= 40.0, p
    assert abs(psnr(got, tgt) - psnr(oout.detach(), tgt)) <= 0.05
    assert abs(float(loss) - float(oloss)) <= 2e-2 * abs(float(oloss)), (float(loss), float(oloss))

    norms = {k: float(osd[k].grad.norm()) for k, _ in m.named_parameters()}
    nmax = max(norms.values())
    dot = gg = ww = 0.0
    worst = worst_spy = (0.0, None)
    for k, prm in m.named_parameters():
        assert prm.grad is not None and torch.isfinite(prm.grad).all(), k
        g, w = prm.grad.float().cpu().double(), osd[k].grad.double()
        dot += float((g * w).sum()); gg += float((g * g).sum()); ww += float((w * w).sum())
        if norms[k] >= 1e-3 * nmax:
            rel = float((g - w).norm()) / norms[k]
            if k.startswith("spynet."):
                if rel > worst_spy[0]:
                    worst_spy = (rel, k)
            elif rel > worst[0]:
                worst = (rel, k)
    cos = dot / (gg ** 0.5 * ww ** 0.5)
    print(f"bench batch gradients: cosine {cos:.5f}, worst relative L2 {worst[0]:.4f} at {worst[1]}; SPyNet: {worst_spy[0]:.4f} at {worst_spy[1]}")
    assert cos >= 0.999, cos
    assert worst[0] <= 0.04, worst
    assert worst_spy[0] <= SPYNET_BF16_BOUND, worst_spy


class _RoundBF16(torch.autograd.Function):
    """x rounded to bf16 on the way forward, its gradient rounded to bf16 on the way back: the storage format of every tensor the bf16 chain
    writes (activations and activation gradients), so that an fp32 autograd run of the oracle chain EMULATES the product's rounding points."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


def _smooth_gradient(shape):
    """An output gradient with spatial structure (5x5 box-filtered noise + a per-channel offset, rounded to bf16), like the gradients the
    chain sees inside the model.  With WHITE noise every weight gradient is a cancelling sum of 32 768 uncorrelated products whose value is
    the size of its own rounding noise (measured: 0.075 relative L2 vs fp32 where the model-level test measures <= 0.024)."""
    import torch.nn.functional as F
    from oracle import recipe as R
    n, h, w, c = shape
    g = R.seeded((n, c, h, w), 743)
    g = F.avg_pool2d(g, 5, 1, 2) * 5.0 + 0.3 * R.seeded((1, c, 1, 1), 744)
    return g.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).float()


def _emulated_bf16_chain(osd, p, x, nblk, r):
    """O.resblocks (models/trajectory.py:16-52, 165-221) with the product's rounding points: every convolution accumulates in fp32 and its
    epilogue result (bias, activation, r-scaled residual add) is stored as bf16; in the backward every activation gradient is stored as bf16
    (the ReLU mask is applied before the rounding, as the data-gradient kernel's epilogue does)."""
    import torch.nn.functional as F
    from oracle import vmg_oracle as O
    rnd = _RoundBF16.apply
    y = rnd(F.leaky_relu(O.conv_nhwc(x, osd[f"{p}main.0.weight"], osd[f"{p}main.0.bias"], 1), 0.1))
    for k in range(nblk):
        q = f"{p}main.2.{k}."
        t = F.relu(rnd(O.conv_nhwc(y, osd[f"{q}conv1.weight"], osd[f"{q}conv1.bias"], 1)))
        y = rnd(y + O.conv_nhwc(t, osd[f"{q}conv2.weight"], osd[f"{q}conv2.bias"], 1) * r)
    return y


@pytest.mark.parametrize("fp8", [False, True])
def test_residual_chain_at_the_bench_size_vs_oracle(fp8):
    """ResidualBlocksWithInputConv(288, 144, 15) on (8, 64, 64) pixels = M 32 768, bf16, forward + backward through ONE vmg_resblock_chain_fwd /
    _bwd call each over the weight-streaming kernel (fp8: vmg_resblock_chain_fwd_q8 for the block convolutions of the forward) vs the oracle chain
    (O.resblocks, models/trajectory.py:16-52), output gradient = _smooth_gradient.  Two comparisons for bf16:
      (1) against the fp32 oracle with the product's bf16 ROUNDING POINTS emulated (_emulated_bf16_chain): what is left is summation order
          and the rounding flips it causes (an element that lands on the other side of a bf16 tie moves by 2^-8 of itself) -- output <= 5e-3,
          input gradient <= 2e-2, every parameter gradient <= 3e-2 relative L2;
      (2) against the plain fp32 oracle, the price of bf16 storage through 31 convolutions and 60 data-gradient convolutions: output <= 1e-2,
          input gradient <= 3e-2, parameter gradients cosine >= 0.998.
    fp8 (stated in tests/test_fp8_gpu.py): output <= 3e-2, input-gradient cosine >= 0.99, parameters >= 0.97 vs the plain fp32 oracle."""
    from oracle import recipe as R, vmg_oracle as O
    from vmg_amd import functional as FH
    from vmg_amd.model import ResidualBlocksWithInputConv
    C_, nblk, n, h, w_ = 144, 15, 8, 64, 64
    m = ResidualBlocksWithInputConv(2 * C_, C_, nblk, 0.1).cuda()
    sd = {k: R.seeded(tuple(v.shape), 730 + i, (v.shape[1] * 9) ** -0.5 if v.dim() == 4 else 0.05) for i, (k, v) in enumerate(m.state_dict().items())}
    # (weights pre-rounded to bf16 on both sides: the comparison then measures the kernels' arithmetic, not the cast of the parameters)
    sd = {k: (v.to(torch.bfloat16).float() if v.dim() == 4 else v) for k, v in sd.items()}
    m.load_state_dict(sd)
    a, b = R.seeded((n, h, w_, C_), 741).to(torch.bfloat16), R.seeded((n, h, w_, C_), 742).to(torch.bfloat16)
    names = sorted("r." + k for k in sd)

    def oracle(emulate):
        osd = {("r." + k): v.clone().requires_grad_(True) for k, v in sd.items()}
        xo = torch.cat([a, b], -1).float().requires_grad_(True)
        out = _emulated_bf16_chain(osd, "r.", xo, nblk, 0.1) if emulate else O.resblocks(osd, "r.", xo, nblk, 0.1)
        go_ = _smooth_gradient(tuple(out.shape))
        gr = torch.autograd.grad(out, [xo] + [osd[k] for k in names], go_)
        return out.detach(), go_, gr

    want, go, wg = oracle(False)
    ad, bd = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    before = _chain_counts()
    q8_before = FH.FP8_STATS["chains"]
    FH.set_fp8_chains(fp8)
    try:
        got = m([ad, bd])
        got.backward(go.cuda().to(torch.bfloat16))
    finally:
        FH.set_fp8_chains(False)
    after = _chain_counts()
    if fp8:
        assert FH.FP8_STATS["chains"] == q8_before + 1
    else:
        assert _delta(before, after, "fwd", 3) == 1
    assert _delta(before, after, "bwd", 3) == 1
    params = dict(m.named_parameters())
    gx = torch.cat([ad.grad, bd.grad], -1).float().cpu()

    def compare(ref_out, ref_g):
        rel = float((got.float().cpu().double() - ref_out.double()).norm() / ref_out.double().norm())
        relx = float((gx.double() - ref_g[0].double()).norm() / ref_g[0].double().norm())
        cosx = float((gx.double() * ref_g[0].double()).sum() / (gx.double().norm() * ref_g[0].double().norm()))
        per = []
        for k, gw in zip(names, ref_g[1:]):
            g = params[k[2:]].grad.cpu().double()
            gw = gw.double()
            per.append((float((g - gw).norm() / (gw.norm() + 1e-30)), float((g * gw).sum() / (g.norm() * gw.norm() + 1e-30)), k))
        return rel, relx, cosx, per

    rel, relx, cosx, per = compare(want, wg)
    worst_rel, worst_cos = max(p[0] for p in per), min(p[1] for p in per)
    print(f"chain at M = 32768 ({'fp8' if fp8 else 'bf16'}) vs the fp32 oracle: output rel L2 {rel:.4f}, input gradient rel L2 {relx:.4f} (cos {cosx:.5f}), "
          f"parameter gradients worst rel L2 {worst_rel:.4f} / worst cosine {worst_cos:.5f}; worst five: "
          + ", ".join(f"{k} {e:.4f}" for e, _, k in sorted(per, reverse=True)[:5]))
    if fp8:
        assert rel <= 3e-2 and cosx >= 0.99 and worst_cos >= 0.97, (rel, cosx, worst_cos)
        return
    assert rel <= 1e-2 and relx <= 3e-2 and worst_cos >= 0.998, (rel, relx, worst_cos)
    want_e, _, wg_e = oracle(True)
    rel, relx, cosx, per = compare(want_e, wg_e)
    worst_rel = max(p[0] for p in per)
    print(f"chain at M = 32768 (bf16) vs the oracle with bf16 rounding points: output rel L2 {rel:.5f}, input gradient rel L2 {relx:.5f}, "
          f"parameter gradients worst rel L2 {worst_rel:.5f} at {max(per)[2]}")
    assert rel <= 5e-3 and relx <= 2e-2 and worst_rel <= 3e-2, (rel, relx, worst_rel)


def test_inference_tile_size_eval_call_vs_oracle():
    """The inference route's network call (bench.py --workload infer feeds (1, 50, 3, 128, 128) tiles: M = 2 * 128 * 128 = 32 768 pixels per chain
    launch, the weight-streaming route again): a (1, 5, 3, 128, 128) eval call in bf16 vs the oracle -- PSNR >= 40 dB (stated bf16 bound) -- and
    in fp32 within the whole-model bound of tests/test_model_gpu.py (max |d| <= 2e-3)."""
    from oracle import cases as C
    from oracle import recipe as R
    from oracle import vmg_oracle as O
    from tests.util import build_product, psnr
    shapes, _ = C.load_fixture(os.path.join(GOLD, "vmg_reds_few_cfg1.npz"))
    chunk_of, window_of = R.vmg_chunk_lookup(C.cfg_reds_few(T=5))
    sd = R.recipe_state_dict(shapes, 0, chunk_of, window_of)
    x = R.synthetic_clip(1, 5, 128, 128, 97)
    with torch.no_grad():
        want = O.vmg_forward({k: v.clone() for k, v in sd.items()}, C.cfg_reds_few(T=5), x)
    before = _chain_counts()
    for dt in (torch.bfloat16, torch.float32):
        m = build_product(C.cfg_reds_few(T=5), dt)
        m.load_state_dict(sd)
        m.eval()
        with torch.no_grad():
            got = m(x.cuda()).float().cpu()
        if dt == torch.bfloat16:
            after = _chain_counts()
            assert _delta(before, after, "fwd", 3) == 2 * 5
            p = psnr(got, want)
            print(f"128x128 eval call: PSNR(hip bf16, oracle) = {p:.2f} dB")
            assert p >= 40.0, p
        else:
            err = float((got - want).abs().max())
            print(f"128x128 eval call: fp32 max |hip - oracle| = {err:.2e}")
            assert err <= 2e-3, err
