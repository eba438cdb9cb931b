"""Uninitialised-read hunt: fill the caching allocator's free blocks with NaN bit patterns, then run the model eagerly.  Any kernel that reads
memory nobody wrote (torch.empty outputs, workspace tails, padded channels) and lets it reach a result shows up as a NaN."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import cases as C
from tests.util import build_product
from vmg_amd import functional as FH
from vmg_amd.data import synthetic_clip, synthetic_target

def poison(total_mb=6000):
    # blocks of many sizes, so that both the small and the large pools hold poisoned free blocks
    bufs = []
    for sz in [1 << k for k in range(9, 28)]:
        reps = max(1, min(64, (total_mb << 20) // 19 // (sz * 4)))
        for _ in range(reps):
            bufs.append(torch.full((sz,), float("nan"), device="cuda"))
    torch.cuda.synchronize()
    del bufs

which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
dt = torch.float32 if (len(sys.argv) < 3 or sys.argv[2] == "fp32") else torch.bfloat16
mode = sys.argv[3] if len(sys.argv) > 3 else "deferred"
if which == "tiny":
    cfg = C.cfg_tiny_few(3, is_train=False); fx = "vmg_tiny_few"; case = C.CASES[fx]
    shapes, _ = C.load_fixture(f"tests/golden/{fx}.npz"); sd = C.case_state_dict(case, shapes)
    x = synthetic_clip(1, 3, 64, 64, seed=75, device="cuda")
else:
    from oracle import recipe as R
    cfg = C.cfg_reds_few(T=5) if which == "few" else C.cfg_reds_full(T=3)
    shapes, _ = C.load_fixture("tests/golden/%s.npz" % ("vmg_reds_few_cfg1" if which == "few" else "vmg_reds_full"))
    ck, wk = R.vmg_chunk_lookup(cfg); sd = R.recipe_state_dict(shapes, 0, ck, wk)
    x = synthetic_clip(2 if which == "few" else 1, cfg.num_frames, 64, 64, seed=75, device="cuda")
y = synthetic_target(x)
FH.set_wgrad_mode(mode)
res = []
for poisoned in (False, True, True):
    m = build_product(cfg, dt)
    m.load_state_dict(sd)
    m.train()
    if poisoned:
        poison()
    found = []
    def watch(name):
        def h(g):
            if not torch.isfinite(g).all() and len(found) < 12:
                found.append(name)
        return h
    # gradient taps inside SPyNet
    spy = m.spynet
    orig = spy.compute_flow
    def compute_flow(ref, supp, spy=spy):
        from vmg_amd import kernels as K
        n, h, w, _ = ref.shape
        d = ref.dtype
        refs, supps = [ref], [supp]
        with torch.no_grad():
            for _ in range(5):
                refs.append(K.avgpool2(refs[-1])); supps.append(K.avgpool2(supps[-1]))
        refs, supps = refs[::-1], supps[::-1]
        flow = torch.zeros(n, h // 32, w // 32, 2, dtype=torch.float32, device=ref.device)
        for level in range(6):
            up = flow if level == 0 else FH.upsample2x_flow(flow, 2.0)
            if up.requires_grad: up.register_hook(watch(f"L{level}.up"))
            warped = FH.grid_sample_flow(supps[level], up, "bilinear", "border")
            if warped.requires_grad: warped.register_hook(watch(f"L{level}.warped"))
            x8 = torch.cat([refs[level][..., :3], warped[..., :3], up.to(d)], -1)
            if x8.requires_grad: x8.register_hook(watch(f"L{level}.x8"))
            r = spy.basic_module[level]([x8])
            r.register_hook(watch(f"L{level}.res"))
            flow = up + r.float()
            flow.register_hook(watch(f"L{level}.flow"))
        return flow
    spy.compute_flow = compute_flow
    out = m(x)
    loss = (out.float() - y).square().mean()
    loss.backward()
    FH.flush_deferred_wgrads()
    torch.cuda.synchronize()
    bad = [k for k, p in m.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    print(f"poisoned={poisoned} {which} {dt} {mode}: loss {float(loss):.6f} out finite {bool(torch.isfinite(out).all())} non-finite grads {len(bad)} {bad[:4]}  first NaN taps (backward order): {found}")
    res.append((out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    del m
print("forward bit-equal clean vs poisoned:", torch.equal(res[0][0], res[1][0]))
worst = max((float((res[0][1][k] - res[1][1][k]).abs().max() / (res[0][1][k].abs().max() + 1e-30)), k) for k in res[0][1] if torch.isfinite(res[1][1][k]).all())
print("largest relative gradient change clean vs poisoned among finite tensors:", worst)
