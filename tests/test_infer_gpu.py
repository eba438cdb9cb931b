"""Sliding-window inference harness (vmg_amd.infer, HIP accumulate / finalize kernels through the C-ABI) vs the CPU oracle
(oracle/infer_oracle.py, pinned by the reference's tools/Tester.py through tests/golden/infer_*.npz).

The window arithmetic is integer work and the accumulators add the same fp32 numbers in the same order, so with the tile
outputs REPLAYED from the oracle's calls the canvases must be bit-exact.  The end-to-end case runs the tiny VMG through the
temporal + spatial windows (four stateful network calls) against the numbers the unmodified reference produced (fp32
tolerance of the whole-model tests: 2e-3)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


class _Recorder:
    """Wraps the oracle-side model: records every call's output so that the GPU side can replay the identical numbers."""

    def __init__(self, model):
        self.model, self.outs = model, []

    def __call__(self, x):
        o = self.model(x)
        self.outs.append(o.clone())
        return o


class _Replay:
    def __init__(self, outs, shapes_check=True):
        self.outs, self.i = outs, 0

    def __call__(self, x):
        o = self.outs[self.i]
        assert tuple(o.shape[-2:]) == (4 * x.shape[-2], 4 * x.shape[-1]) and x.is_cuda
        self.i += 1
        return o.cuda()


@pytest.mark.parametrize("shape,spatial,ov", [((1, 3, 3, 40, 52), [16, 20], 6), ((1, 3, 3, 40, 52), [16, 20], 5), ((2, 2, 3, 33, 47), [16, 16], 7),
                                              ((1, 2, 3, 16, 20), [16, 20], 6), ((1, 1, 3, 64, 64), [24, 40], 0)])
def test_test_image_bit_exact(shape, spatial, ov):
    from oracle import infer_oracle as IO
    from oracle import recipe as R
    from vmg_amd import infer
    x = R.seeded(shape, 201, 0.3) + 0.5
    rec = _Recorder(IO.fake_sr_model())
    want = IO.test_image(rec, x, spatial, ov, 4)
    rep = _Replay(rec.outs)
    got = infer.test_image(rep, x.cuda(), spatial, ov, 4)
    assert rep.i == len(rec.outs)
    g = got.cpu()
    assert torch.equal(torch.isnan(g), torch.isnan(want))  # (overlap 0 with neighbours: the reference's 0/0, reproduced)
    assert torch.equal(torch.nan_to_num(g), torch.nan_to_num(want))


@pytest.mark.parametrize("T,nf,of,spatial,ov", [(11, 5, 2, [16, 16], 4), (11, 5, 3, None, None), (11, 4, 0, [16, 16], 4), (3, 5, 2, None, None), (9, 3, 1, None, None)])
def test_test_clips_bit_exact(T, nf, of, spatial, ov):
    from oracle import infer_oracle as IO
    from oracle import recipe as R
    from vmg_amd import infer
    x = R.seeded((1, T, 3, 24, 28), 202, 0.3) + 0.5
    nf = min(nf, T)
    rec = _Recorder(IO.fake_sr_model())
    want = IO.test_clips(rec, x, nf, of, spatial, ov, 4)
    rep = _Replay(rec.outs)
    got = infer.test_clips(rep, x.cuda(), nf, of, spatial, ov, 4)
    assert rep.i == len(rec.outs)
    assert torch.equal(got.cpu(), want)


def test_test_clips_max_and_uint8():
    from oracle import cases as C
    from oracle import infer_oracle as IO
    from vmg_amd import infer
    inp = C.CASES["infer_clips_max"]["inputs"]()
    rec = _Recorder(IO.fake_sr_model())
    want = IO.test_clips_max(rec, inp["x"], inp["hr"], 4, 2, None, None, 4)
    got = infer.test_clips_max(_Replay(rec.outs), inp["x"].cuda(), inp["hr"].cuda(), 4, 2, None, None, 4)
    assert torch.equal(got.cpu(), want)
    assert np.array_equal(infer.to_uint8(got), IO.to_uint8(want))
    # and against the reference's own numbers
    _, ref = C.load_fixture(os.path.join(GOLD, "infer_clips_max.npz"))
    assert float(np.abs(C.subsample(got.cpu()) - ref[0]["sub"]).max()) <= 1e-6
    assert float(np.abs(C.subsample(torch.from_numpy(infer.to_uint8(got).astype(np.float32))) - ref[1]["sub"]).max()) == 0.0
    # round-half-even on exact ties: k/510 * 255 = k/2
    ties = ((torch.arange(0, 7 * 3 * 73 * 2) % 511).to(torch.float32) / 510.0).reshape(1, 7, 3, 73, 2)
    assert np.array_equal(infer.to_uint8(ties.cuda()), IO.to_uint8(ties))


def test_vmg_sliding_windows_match_reference():
    """Tiny VMG (fp32) through temporal windows 3/1 and spatial tiles 64/8 on a (1,5,3,72,64) clip: four stateful calls."""
    from oracle import cases as C
    from tests.util import build_product
    from vmg_amd import infer
    case = C.CASES["infer_vmg_clips"]
    shapes, ref = C.load_fixture(os.path.join(GOLD, "infer_vmg_clips.npz"))
    sd = C.case_state_dict(case, shapes)
    m = build_product(case["cfg"], torch.float32)
    m.load_state_dict(sd, strict=True)
    m.eval()
    x = case["inputs"]()["x"]
    got = infer.test_clips(m, x.cuda(), 3, 1, [64, 64], 8, 4).cpu()
    assert tuple(got.shape) == ref[0]["shape"]
    assert float(np.abs(C.subsample(got) - ref[0]["sub"]).max()) <= 2e-3
    want = case["run"]({k: v.clone() for k, v in sd.items()}, case["inputs"]())[0]
    assert float((got - want).abs().max()) <= 2e-3


def test_infer_refuses_cpu():
    from vmg_amd import hip, infer
    with pytest.raises(hip.HipError):
        infer.test_image(lambda x: x, torch.zeros(1, 1, 3, 16, 16), [16, 16], 4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_graphed_model_gives_the_same_bits_and_keeps_the_call_count(dtype):
    """infer.GraphedModel: the network call replayed from a captured graph.  The sliding-window result must equal the eager one bit for bit -- including
    the statefulness: the warm-up calls the capture needs must not count as calls (the mixer weights are restored), and every replay must
    decay the weights once like an eager call does."""
    from oracle import cases as C
    from tests.util import build_product
    from vmg_amd import infer
    case = C.CASES["infer_vmg_clips"]
    shapes, _ = C.load_fixture(os.path.join(GOLD, "infer_vmg_clips.npz"))
    sd = C.case_state_dict(case, shapes)
    x = case["inputs"]()["x"].cuda()
    res = []
    for graphed in (False, True):
        m = build_product(case["cfg"], dtype)
        m.load_state_dict(sd, strict=True)
        m.eval()
        net = infer.GraphedModel(m) if graphed else m
        out = infer.test_clips(net, x, 3, 1, [64, 64], 8, 4)
        res.append((out.float().cpu(), {k: v.clone() for k, v in m.state_dict().items() if "mlp_h.0.weight" in k or "mlp_w.0.weight" in k}))
    assert torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k  # the same number of decays has been applied


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_graphed_model_two_shapes_a_b_a(dtype):
    """GraphedModel keeps one hipGraph per input shape, and every graph has baked in the addresses of the one-launch repack
    plan's entry table and of the weight packs it rewrites.  A call at a second shape creates new packs (other tilings) and rebuilds the plan:
    the first graph must still replay correctly afterwards (the superseded table is retired, not freed; stale packs are rewritten in place).
    Calls A, B, A, B through the wrapper == the same four eager calls on a second model, bit for bit, including the per-call weight decay."""
    from oracle import cases as C
    from oracle import recipe as R
    from tests.util import build_product
    from vmg_amd import infer
    from vmg_amd import functional as FH
    case = C.CASES["infer_vmg_clips"]
    shapes, _ = C.load_fixture(os.path.join(GOLD, "infer_vmg_clips.npz"))
    sd = C.case_state_dict(case, shapes)
    xa = R.synthetic_clip(1, 3, 64, 64, 95).cuda()
    xb = R.synthetic_clip(1, 3, 96, 80, 96).cuda()
    res = []
    for graphed in (False, True):
        FH.clear_pack_cache()
        m = build_product(case["cfg"], dtype)
        m.load_state_dict(sd, strict=True)
        m.eval()
        net = infer.GraphedModel(m) if graphed else m
        outs = []
        with torch.no_grad():
            for x in (xa, xb, xa, xb):
                outs.append(net(x).float().cpu().clone())
                # allocator churn between the calls: a freed plan table / pack buffer would be handed out again here and overwritten
                junk = [torch.full((1 << 12,), 7.0, device="cuda") for _ in range(64)]
                del junk
        res.append(outs)
        if graphed:
            assert len(net.graphs) == 2
    for i, (a, b) in enumerate(zip(*res)):
        assert torch.equal(a, b), f"call {i}: graphed and eager results differ"
