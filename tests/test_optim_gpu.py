"""FlatAdamW (one HIP kernel per parameter group over a flat buffer, through the C-ABI) against torch.optim.AdamW -- which
IS the reference's optimizer (tools/Trainer.py:86-105) -- on the same parameters and gradients, three groups (lr 0, plain,
weight decay), several steps with changing learning rates.  fp32: relative 2e-6 (fused multiply-adds round differently)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(144, 144, 3, 3), (144,), (7,), (33, 5), (144, 288), (1,), (2, 3, 5, 7)]
    return [torch.nn.Parameter(torch.randn(s, generator=g) * 0.1) for s in shapes]


def test_flat_adamw_matches_torch_adamw():
    from vmg_amd.train import FlatAdamW
    ref_p = _params(1)
    my_p = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_p]

    def groups(ps):
        return [{"params": ps[:2], "lr": 0.0}, {"params": ps[2:5]}, {"params": ps[5:], "weight_decay": 0.05}]
    ref = torch.optim.AdamW(groups(ref_p), lr=2e-4, betas=(0.9, 0.99), weight_decay=0.0)
    opt = FlatAdamW(groups(my_p), lr=2e-4, betas=(0.9, 0.99), weight_decay=0.0)
    assert all(p.grad is not None and p.data_ptr() >= opt.p.data_ptr() for p in my_p)
    gen = torch.Generator().manual_seed(2)
    for step in range(6):
        lr = 2e-4 * (1.0 - 0.1 * step)
        ref.param_groups[1]["lr"] = ref.param_groups[2]["lr"] = lr
        opt.param_groups[1]["lr"] = opt.param_groups[2]["lr"] = lr
        if step == 3:
            ref.param_groups[0]["lr"] = opt.param_groups[0]["lr"] = 1e-6  # the SPyNet group wakes up (eta_min of the cosine schedule)
        for rp, mp in zip(ref_p, my_p):
            gr = torch.randn(rp.shape, generator=gen) * 0.01
            rp.grad = gr.clone()
            mp.grad.copy_(gr.cuda())  # gradients are persistent views of the flat buffer: written in place, as autograd does
        ref.step()
        opt.step()
        opt.zero_grad()
        assert float(opt.g.abs().max()) == 0.0
    for rp, mp in zip(ref_p, my_p):
        err = float((mp.detach().cpu() - rp.detach()).abs().max())
        assert err <= 2e-6 * max(1.0, float(rp.detach().abs().max())), f"{tuple(rp.shape)}: {err}"
    # moments too (exp_avg of the first tensor of group 1)
    st = ref.state[ref_p[2]]
    o = opt.offsets[2]
    assert float((opt.m[o:o + 7].cpu() - st["exp_avg"]).abs().max()) <= 1e-8
    assert float((opt.v[o:o + 7].cpu() - st["exp_avg_sq"]).abs().max()) <= 1e-10


def test_train_step_uses_flat_buffers_and_repacks_weights():
    """One TrainStep on the tiny model: parameters live in the flat buffer, every parameter moved, and the conv weight packs
    follow the optimizer's in-place update (weight epoch), i.e. a second step sees the new weights."""
    from oracle import cases as C
    from tests.util import build_product
    from vmg_amd.train import FlatAdamW, TrainStep
    from vmg_amd.data import synthetic_clip, synthetic_target
    cfg = C.cfg_tiny_few(3, is_train=True)
    m = build_product(cfg, torch.bfloat16).train()
    ts = TrainStep(m, lr=1e-3)
    assert isinstance(ts.opt, FlatAdamW)
    lo, hi = ts.opt.p.data_ptr(), ts.opt.p.data_ptr() + 4 * ts.opt.n
    assert all(lo <= p.data_ptr() < hi for p in m.parameters())
    x = synthetic_clip(1, 3, 64, 64, seed=5, device="cuda")
    y = synthetic_target(x)
    before = ts.opt.p.clone()
    with torch.no_grad():
        out0 = m.eval()(x).float().clone()
    m.train()
    l1 = float(ts(x, y))
    moved = (ts.opt.p != before)
    spy_n = sum(p.numel() for p in m.spynet.parameters())
    assert int(moved[ts.opt.groups[1]["start"]:].sum()) > 0.9 * (ts.opt.n - spy_n)  # group 0 (SPyNet) has lr 0
    with torch.no_grad():
        out1 = m.eval()(x).float()
    assert float((out1 - out0).abs().max()) > 0  # the forward runs on the UPDATED weights (packs were invalidated)
    m.train()
    l2 = float(ts(x, y))
    assert l1 == l1 and l2 == l2


@pytest.mark.parametrize("shape", [(1, 2, 3, 32, 32), (2, 3, 3, 65, 47), (1, 1, 3, 256, 256), (1, 1, 3, 5, 3)])
def test_charbonnier_edge_loss_kernels(shape):
    """csrc/loss.hip (Laplacian pyramid of x - y with replicate padding and its adjoint) against the oracle's loss -- pinned
    by the reference's CharbonnierLoss fixture -- in value and gradient, odd sizes and the fixture's own inputs."""
    import os
    import numpy as np
    from oracle import cases as C
    from oracle import recipe as R
    from oracle import vmg_oracle as O
    from vmg_amd.train import charbonnier_edge_loss_hip
    x, y = R.seeded(shape, 301, 0.3) + 0.5, R.seeded(shape, 302, 0.3) + 0.5
    xo = x.clone().requires_grad_(True)
    lo = O.charbonnier_edge_loss(xo, y)
    lo.backward()
    xg = x.cuda().requires_grad_(True)
    lg = charbonnier_edge_loss_hip(xg, y.cuda())
    lg.backward()
    assert abs(float(lg) - float(lo)) <= 2e-6 * max(1.0, abs(float(lo)))
    gmax = float(xo.grad.abs().max())
    assert float((xg.grad.cpu() - xo.grad).abs().max()) <= 2e-5 * gmax + 1e-12
    if shape == (1, 2, 3, 32, 32):  # the reference fixture's case
        inp = C.CASES["loss"]["inputs"]()
        _, ref = C.load_fixture(os.path.join(os.path.dirname(__file__), "golden", "loss.npz"))
        got = charbonnier_edge_loss_hip(inp["x"].cuda(), inp["y"].cuda())
        assert abs(float(got) - float(ref[0]["sub"][0])) <= 2e-6 * max(1.0, abs(float(ref[0]["sub"][0])))


def test_replay_of_a_captured_sequence_matches_the_graph():
    """csrc/replay.hip: the kernel / memset nodes of a captured stream are read out of the hipGraph and re-issued with plain launches; the
    result must equal what torch's own graph replay produces (elementwise kernels, a fill, a HIP-library kernel)."""
    import ctypes
    from vmg_amd import hip, kernels as K
    x = torch.randn(4, 16, 16, 32, device="cuda").to(torch.bfloat16)
    y = torch.randn(4, 16, 16, 32, device="cuda").to(torch.bfloat16)
    out = torch.zeros(4, 16, 16, 32, device="cuda", dtype=torch.bfloat16)

    def body():
        t = K.act_backward(x, y, hip.ACT_RELU, 0.0, 1.0)
        u = (t.float() * 2 + 1).to(torch.bfloat16)
        out.zero_()
        out.add_(u)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g):
        body()
    n_ops, n_k = ctypes.c_int(0), ctypes.c_int(0)
    h = hip.lib().vmg_replay_build(ctypes.c_void_p(int(g.raw_cuda_graph())), ctypes.byref(n_ops), ctypes.byref(n_k))
    assert h, hip.lib().vmg_last_error().decode()
    assert n_k.value >= 3 and n_ops.value >= n_k.value
    out.fill_(7)
    x.mul_(-1)  # new inputs: the replay must recompute from them
    hip.check(hip.lib().vmg_replay_run(ctypes.c_void_p(h), 0, n_ops.value, hip.stream_ptr()), "vmg_replay_run")
    torch.cuda.synchronize()
    got = out.clone()
    out.fill_(7)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, out)
    hip.lib().vmg_replay_destroy(ctypes.c_void_p(h))


def test_flat_adamw_state_dict_survives_the_relayout():
    """FlatAdamW.state_dict() is keyed per parameter in construction order: a checkpoint written AFTER relayout() (the data-parallel
    re-layout by gradient-completion order) loads into a fresh optimizer in registration-order layout with every moment on its own
    parameter, and a checkpoint of the fresh layout loads into a re-laid-out optimizer; mismatching parameter lists are refused."""
    from vmg_amd.train import FlatAdamW

    def make(seed=1):
        ps = [torch.nn.Parameter(p.detach().clone().cuda()) for p in _params(seed)]
        return ps, FlatAdamW([{"params": ps[:2], "lr": 0.0}, {"params": ps[2:]}], lr=1e-3)

    pa, a = make()
    gen = torch.Generator().manual_seed(3)
    for _ in range(2):
        for p in pa:
            p.grad.copy_((torch.randn(p.shape, generator=gen) * 0.01).cuda())
        a.step()
    a.relayout([pa[6], pa[3], pa[5], pa[1], pa[0], pa[2], pa[4]])  # permute inside the groups
    for p in pa:
        p.grad.copy_((torch.randn(p.shape, generator=gen) * 0.01).cuda())
    a.step()
    sd = a.state_dict()
    off_a = {id(p): o for p, o in zip(a.params, a.offsets)}
    pb, b = make()
    b.load_state_dict(sd)
    assert b.t == a.t == 3
    off_b = {id(p): o for p, o in zip(b.params, b.offsets)}
    assert [off_a[id(p)] for p in pa] != [off_b[id(p)] for p in pb]  # the two layouts really differ
    for x, y in zip(pa, pb):
        n = x.numel()
        assert torch.equal(a.m[off_a[id(x)]:off_a[id(x)] + n], b.m[off_b[id(y)]:off_b[id(y)] + n])
        assert torch.equal(a.v[off_a[id(x)]:off_a[id(x)] + n], b.v[off_b[id(y)]:off_b[id(y)] + n])
    # and back: the fresh layout's checkpoint into the re-laid-out optimizer; the next steps agree bit for bit
    a.load_state_dict(b.state_dict())
    with torch.no_grad():
        for x, y in zip(pa, pb):
            y.copy_(x)
    for p, q in zip(pa, pb):
        g = (torch.randn(p.shape, generator=gen) * 0.01).cuda()
        p.grad.copy_(g)
        q.grad.copy_(g)
    a.step()
    b.step()
    for x, y in zip(pa, pb):
        assert torch.equal(x.detach(), y.detach())
    ps, c = make()
    bad = dict(sd)
    bad["numel"] = sd["numel"][:-1]
    with pytest.raises(ValueError):
        c.load_state_dict(bad)
    with pytest.raises(ValueError):
        c.load_state_dict({"t": 1, "m": a.m, "v": a.v, "groups": sd["groups"]})  # (round 2's positional format)


@pytest.mark.parametrize("max_norm", [0.05, 1e6])
def test_clip_grad_norm_on_the_flat_buffer(max_norm):
    """vmg_grad_clip_norm against torch.nn.utils.clip_grad_norm_ (what tools/Trainer.py:141-143 calls): total norm, coefficient, clipped
    gradients; a max_norm above the norm leaves the gradients untouched; two runs give the same bits."""
    from vmg_amd.train import FlatAdamW
    ps = [torch.nn.Parameter(p.detach().clone().cuda()) for p in _params(4)]
    opt = FlatAdamW([{"params": ps[:2], "lr": 0.0}, {"params": ps[2:]}], lr=1e-3)
    gen = torch.Generator().manual_seed(5)
    gs = [torch.randn(p.shape, generator=gen) * 0.02 for p in ps]
    refs = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    for r, g in zip(refs, gs):
        r.grad = g.clone()
    want_norm = torch.nn.utils.clip_grad_norm_(refs, max_norm, norm_type=2)
    outs = []
    for _ in range(2):
        for p, g in zip(ps, gs):
            p.grad.copy_(g.cuda())
        res = opt.clip_grad_norm_(max_norm).clone()
        outs.append((res, opt.g.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert abs(float(outs[0][0][0]) - float(want_norm)) <= 1e-6 * float(want_norm)
    for p, r, g in zip(ps, refs, gs):
        assert float((p.grad.cpu() - r.grad).abs().max()) <= 1e-6 * float(g.abs().max())
        if max_norm > 1.0:
            assert torch.equal(p.grad.cpu(), g)


def test_train_step_schedule_clip_and_accumulation():
    """TrainStep with the reference's step pieces (tools/Trainer.py:125-190, 244-272): (a) the learning rates follow LRSchedule after every
    optimizer step -- SPyNet stays frozen until flow_fix and then moves; (b) grad_clip bounds the gradient norm the optimizer sees; (c) two
    accumulated micro-steps (loss / 2 each, update on the second) leave the same gradient in the flat buffer as the sum of two separate
    backward passes halved."""
    from oracle import cases as C
    from tests.util import build_product
    from vmg_amd.train import TrainStep
    from vmg_amd.data import synthetic_clip, synthetic_target
    cfg = C.cfg_tiny_few(3, is_train=False)  # (drop-path 0: the passes are repeatable)
    shapes, _ = C.load_fixture(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "vmg_tiny_few.npz"))
    sd = C.case_state_dict(C.CASES["vmg_tiny_few"], shapes)
    xs = [synthetic_clip(1, 3, 64, 64, seed=70 + i, device="cuda") for i in range(2)]
    ys = [synthetic_target(x) for x in xs]

    m = build_product(cfg, torch.float32)
    m.load_state_dict(sd)
    m.train()
    ts = TrainStep(m, lr=2e-4, schedule=dict(T_period=[40], eta_min=1e-7, flow_fix=1, pre_lr_ratio=0.125, warmup_iter=-1), grad_clip=1e-3)
    spy0 = torch.cat([p.detach().reshape(-1).clone() for p in m.spynet.parameters()])
    seen = []
    ts.grad_hook = lambda t: seen.append(float(t.opt.g.double().square().sum().sqrt()))
    lrs = []
    for it in range(4):
        ts(xs[it % 2], ys[it % 2])
        lrs.append([g["lr"] for g in ts.opt.param_groups])
        spy = torch.cat([p.detach().reshape(-1) for p in m.spynet.parameters()])
        if it <= 1:  # steps 0 and 1 ran with SPyNet's lr 0 (cur_iter <= flow_fix keeps it frozen for the NEXT step too)
            assert torch.equal(spy, spy0), it
    assert not torch.equal(spy, spy0)  # step 3 ran with lr = group 1's lr * 0.125
    want = C.oracle_lr_update(dict(T_period=[40], restarts=None, weights=None, eta_min=1e-7, base=[0.0, 2e-4], flow_fix=1, pre_lr_ratio=0.125,
                                   warmup_iter=-1, reduced_iter=None, steps=4))
    assert lrs == want
    assert all(n <= 1e-3 * (1 + 1e-5) for n in seen), seen  # the optimizer saw clipped gradients
    assert float(ts.grad_norm[0]) > 1e-3 and 0 < float(ts.grad_norm[1]) < 1

    # (c) accumulation
    def fresh():
        mm = build_product(cfg, torch.float32)
        mm.load_state_dict(sd)
        mm.train()
        return mm, TrainStep(mm, lr=0.0)
    m1, t1 = fresh()
    got = {}
    t1.grad_hook = lambda t: got.update({n: p.grad.clone() for n, p in m1.named_parameters()})
    t1(xs[0], ys[0], grad_acc=2, update=False)
    assert t1.iter == 0
    t1(xs[1], ys[1], grad_acc=2, update=True)
    assert t1.iter == 1 and float(t1.opt.g.abs().max()) == 0.0
    single = []
    for i in range(2):
        m2, t2 = fresh()
        for j in range(i):  # T1: every forward call decays the MorphFC mixer weights in place -- micro-step i runs on weights decayed i + 1 times
            with torch.no_grad():
                m2(xs[j])
        g2 = {}
        t2.grad_hook = lambda t, g2=g2, m2=m2: g2.update({n: p.grad.clone() for n, p in m2.named_parameters()})
        t2(xs[i], ys[i])
        single.append(g2)
    gmax = max(float(v.abs().max()) for v in single[0].values())
    for n in got:
        ref = 0.5 * (single[0][n] + single[1][n])
        # (5e-3 of the gradient's scale as in tests/test_grad_gpu.py: LayerNorm's weight gradient is a cancelling sum over 12 288 rows)
        assert float((got[n] - ref).abs().max()) <= 5e-3 * max(float(ref.abs().max()), 1e-3 * gmax), n


def test_captured_step_advances_the_schedule_and_state_dict_resumes():
    """the learning-rate schedule must progress under replay() -- the capture pass and every replay used to leave
    `schedule.step` / `iter` untouched, so warm-up, cosine decay and SPyNet's flow_fix unfreeze never happened in a captured run.  After k eager
    warm-up steps (they are real optimizer steps and count) and N replays the groups' rates are row k + N - 1 of the reference's
    update_learning_rate recursion (oracle_lr_update, pinned by tests/golden/lr_update.npz), SPyNet has moved once cur_iter passed flow_fix,
    and TrainStep.state_dict() -> load_state_dict() resumes iter / schedule / optimizer (tools/Trainer.py:355-365)."""
    from oracle import cases as C
    from tests.util import build_product
    from vmg_amd.train import TrainStep
    from vmg_amd.data import synthetic_clip, synthetic_target
    import os
    cfg = C.cfg_tiny_few(3, is_train=False)
    shapes, _ = C.load_fixture(os.path.join(os.path.dirname(__file__), "golden", "vmg_tiny_few.npz"))
    sd = C.case_state_dict(C.CASES["vmg_tiny_few"], shapes)
    x = synthetic_clip(1, 3, 64, 64, seed=75, device="cuda")
    y = synthetic_target(x)
    sched = dict(T_period=[40], eta_min=1e-7, flow_fix=6, pre_lr_ratio=0.125, warmup_iter=3)
    m = build_product(cfg, torch.float32)
    m.load_state_dict(sd)
    m.train()
    ts = TrainStep(m, lr=2e-4, schedule=dict(sched))
    spy0 = torch.cat([p.detach().reshape(-1).clone() for p in m.spynet.parameters()])
    ts.capture(x, y, warmup=2)
    k = ts.iter
    assert 2 <= k <= 8, k                     # the eager warm-up steps; the capture pass itself is not a step
    assert ts.schedule.epoch == k
    want = C.oracle_lr_update(dict(T_period=[40], restarts=None, weights=None, eta_min=1e-7, base=[0.0, 2e-4], flow_fix=6, pre_lr_ratio=0.125,
                                   warmup_iter=3, reduced_iter=None, steps=k + 6))
    assert [g["lr"] for g in ts.opt.param_groups] == want[k - 1]
    frozen_until = None
    for n in range(6):
        ts(x, y)
        assert ts.iter == k + n + 1
        assert [g["lr"] for g in ts.opt.param_groups] == want[k + n], (n, k)
        torch.cuda.synchronize()
        spy = torch.cat([p.detach().reshape(-1) for p in m.spynet.parameters()])
        assert torch.isfinite(spy).all(), ts.iter
        if torch.equal(spy, spy0):
            frozen_until = ts.iter
        else:
            print(f"iter {ts.iter}: SPyNet moved by {float((spy - spy0).abs().max()):.3e} (lr group 0 was {want[k + n - 1][0]:.3e})")
    # step number i (0-based) runs with the rates update_learning_rate(i - 1) left: SPyNet's is 0 while i - 1 <= flow_fix, i.e. steps 0 .. 7
    assert frozen_until == min(k + 6, 8), (frozen_until, k)
    assert (k + 6 <= 8) or not torch.equal(spy, spy0)

    state = ts.state_dict()
    m2 = build_product(cfg, torch.float32)
    m2.load_state_dict(sd)
    m2.train()
    t2 = TrainStep(m2, lr=2e-4, schedule=dict(sched))
    t2.load_state_dict(state)
    assert t2.iter == ts.iter and t2.schedule.epoch == ts.schedule.epoch and t2.opt.t == ts.opt.t
    assert [g["lr"] for g in t2.opt.param_groups] == [g["lr"] for g in ts.opt.param_groups]
    with pytest.raises(ValueError):
        TrainStep(m2, lr=2e-4).load_state_dict(state)


def test_accumulation_clips_every_micro_step_like_the_reference():
    """tools/Trainer.py:179-180: with gradient accumulation the reference clips after EVERY micro-step's backward, so the partial sum is
    rescaled before the next micro-step adds to it.  Two micro-steps with a tight max_norm: the buffer after micro-step 1 has norm <= max_norm,
    and the final gradient equals clip(clip(g1) + g2) built from two separately measured gradients."""
    from oracle import cases as C
    from tests.util import build_product
    from vmg_amd.train import TrainStep
    from vmg_amd.data import synthetic_clip, synthetic_target
    import os
    cfg = C.cfg_tiny_few(3, is_train=False)
    shapes, _ = C.load_fixture(os.path.join(os.path.dirname(__file__), "golden", "vmg_tiny_few.npz"))
    sd = C.case_state_dict(C.CASES["vmg_tiny_few"], shapes)
    xs = [synthetic_clip(1, 3, 64, 64, seed=80 + i, device="cuda") for i in range(2)]
    ys = [synthetic_target(x) for x in xs]

    def fresh(clip):
        mm = build_product(cfg, torch.float32)
        mm.load_state_dict(sd)
        mm.train()
        return mm, TrainStep(mm, lr=0.0, grad_clip=clip)
    max_norm = 1e-3
    m1, t1 = fresh(max_norm)
    t1(xs[0], ys[0], grad_acc=2, update=False)
    n1 = float(t1.opt.g.double().square().sum().sqrt())
    assert n1 <= max_norm * (1 + 1e-5) and float(t1.grad_norm[0]) > max_norm  # micro-step 1 was clipped
    got = {}
    t1.grad_hook = lambda t: got.update({n: p.grad.clone() for n, p in m1.named_parameters()})
    t1(xs[1], ys[1], grad_acc=2, update=True)
    raw = []
    for i in range(2):
        m2, t2 = fresh(None)
        for j in range(i):
            with torch.no_grad():
                m2(xs[j])  # (T1: micro-step i runs on mixer weights decayed i + 1 times)
        g2 = {}
        t2.grad_hook = lambda t, g2=g2, m2=m2: g2.update({n: p.grad.clone() for n, p in m2.named_parameters()})
        t2(xs[i], ys[i], grad_acc=2, update=True)
        raw.append(g2)

    def clip(gs):
        tot = float(torch.sqrt(sum(v.double().square().sum() for v in gs.values())))
        c = min(1.0, max_norm / (tot + 1e-6))
        return {n: v * c for n, v in gs.items()}
    step1 = clip(raw[0])
    want = clip({n: step1[n] + raw[1][n] for n in step1})
    gmax = max(float(v.abs().max()) for v in want.values())
    for n in got:
        assert float((got[n] - want[n]).abs().max()) <= 5e-3 * max(float(want[n].abs().max()), 1e-3 * gmax), n


def test_replayed_steps_survive_allocator_churn():
    """A captured step replayed 10 times while the caching allocator hands the same free blocks to NaN-filled tensors between replays: SPyNet, whose
    learning rate is 0, must stay bit-identical (0 x NaN is NaN: any non-finite gradient shows), the loss finite, and no tensor allocated between
    replays may be overwritten by the graph (a node writing through the address of memory that was freed after the capture).  This is the scenario in
    which round 3's `hipMemsetAsync` node and the freed pack-plan tables failed."""
    import os
    from oracle import cases as C
    from tests.util import build_product
    from vmg_amd.train import TrainStep
    from vmg_amd.data import synthetic_clip, synthetic_target
    cfg = C.cfg_tiny_few(3, is_train=False)
    shapes, _ = C.load_fixture(os.path.join(os.path.dirname(__file__), "golden", "vmg_tiny_few.npz"))
    sd = C.case_state_dict(C.CASES["vmg_tiny_few"], shapes)
    x = synthetic_clip(1, 3, 64, 64, seed=76, device="cuda")
    y = synthetic_target(x)
    m = build_product(cfg, torch.float32)
    m.load_state_dict(sd)
    m.train()
    ts = TrainStep(m, lr=2e-4, spynet_lr=0.0)
    spy0 = torch.cat([p.detach().reshape(-1).clone() for p in m.spynet.parameters()])
    ts.capture(x, y, warmup=2)
    canaries = []
    for n in range(10):
        loss = ts(x, y)
        torch.cuda.synchronize()
        assert torch.isfinite(loss), n
        spy = torch.cat([p.detach().reshape(-1) for p in m.spynet.parameters()])
        assert torch.equal(spy, spy0), f"replay {n}: SPyNet (lr 0) changed by {float((spy - spy0).abs().max())}"
        for c in canaries:
            assert bool(torch.isnan(c).all()), f"replay {n}: the graph wrote into memory allocated after the capture"
        canaries = [torch.full((sz,), float("nan"), device="cuda") for sz in (256, 4096, 65536, 1 << 20) for _ in range(6)]
