"""Generate tests/golden/*.npz by running the UNMODIFIED reference (build container only).

    python -m oracle.gen_golden [case ...]

TEST INFRASTRUCTURE ONLY.  Puts oracle/_standins (tiny stand-ins for the third-party names the reference
imports but the image lacks: timm DropPath/trunc_normal_/to_2tuple, mmcv ConvModule = Conv2d+ReLU,
torchvision/thop names that are never called on the configured path) and /root/reference on sys.path,
builds the reference's own modules, loads weights-by-recipe (oracle/recipe.py), runs them on the seeded
inputs of oracle/cases.py and stores a strided subsample + float64 checksums of every output.

Nothing from /root/reference is copied: fixtures hold only inputs-by-seed metadata (state-dict shapes) and
output numbers.  The construction-time buffers rebuilt by recipe.buffer_tensor are asserted equal to the
reference's own buffers here, which pins decay_gamma / decay_v / relative_position_index.
"""
from __future__ import annotations

import os
import sys
import warnings

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")


def _import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference not present: fixtures can only be regenerated in the build container")
    sys.path.insert(0, os.path.join(HERE, "_standins"))
    sys.path.insert(1, REF)
    warnings.filterwarnings("ignore")
    import models.function as Fn  # noqa
    import models.layers as L  # noqa
    import models.swin_3d as S3  # noqa
    import models.trajectory as Tj  # noqa
    import models.vmg as V  # noqa
    # utils/__init__.py pulls in cv2 (absent, ordinary ModuleNotFoundError); utils/loss.py itself only needs torch
    import importlib.util
    spec = importlib.util.spec_from_file_location('vmg_ref_loss', os.path.join(REF, 'utils', 'loss.py'))
    Ls = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(Ls)
    return Fn, L, S3, Tj, V, Ls


def _load_recipe(module, case):
    from . import cases as C
    from . import recipe as R
    own = {k: v.clone() for k, v in module.state_dict().items()}
    shapes = {k: list(v.shape) for k, v in own.items()}
    sd = C.case_state_dict(case, shapes)
    for k in sd:
        if R.is_buffer(k):
            ref_buf = own[k]
            assert ref_buf.shape == sd[k].shape, k
            if ref_buf.dtype.is_floating_point:
                assert torch.allclose(ref_buf, sd[k].to(ref_buf.dtype), rtol=0, atol=2e-7), (k, (ref_buf - sd[k]).abs().max())
            else:
                assert torch.equal(ref_buf, sd[k].to(ref_buf.dtype)), k
            sd[k] = ref_buf  # keep the reference's own bits
    module.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    module.eval()
    return shapes


def _vmg_from_cfg(V, cfg):
    m = V.VMG(embed_dim=list(cfg.embed_dim), depths=list(cfg.depths), num_heads=list(cfg.num_heads), num_frames=cfg.num_frames,
              window_sizes=[list(w) for w in cfg.window_sizes], mdsc=cfg.mdsc, if_concat=False, mlp_ratio=cfg.mlp_ratio,
              n_groups=cfg.n_groups, spynet_pretrained=None, image_size=list(cfg.image_size), is_train=cfg.is_train,
              ltam=True, traj_win=list(cfg.traj_win), traj_keyframes_n=list(cfg.traj_keyframes_n), traj_heads=list(cfg.traj_heads),
              temporal_type=list(cfg.temporal_type), temporal_empty=cfg.temporal_empty, traj_res_n=list(cfg.traj_res_n),
              spatial_type=list(cfg.spatial_type), flow_smooth=cfg.flow_smooth, smooth_region_range=cfg.smooth_region_range,
              retention_decay=True, non_linear=True, gating=True, symm=True, symm_act="tanh", relu_scale=True,
              relu_scale_norm=False, ffn_type=cfg.ffn_type, mixer_type=["mlps"] * cfg.num_enc_layers,
              mixer_n=[None] * cfg.num_enc_layers, r_scaling=cfg.r_scaling, chunk_ratios=list(cfg.chunk_ratios),
              traj_mode="wins", twins=list(cfg.twins), traj_scale=cfg.traj_scale, traj_refine=None, m_scaling=cfg.m_scaling,
              if_local_fuse=cfg.if_local_fuse, channel_mixer=cfg.channel_mixer,
              deform_groups=[8] * cfg.num_enc_layers, max_residual_scale=[1] * cfg.num_enc_layers)
    m.spynet = V.SPyNet(None)  # attach a random-init SPyNet after construction
    return m


def reference_side(name, case, mods):
    """Returns (shapes, outputs) of the reference for one case."""
    import torch.nn as nn
    Fn, L, S3, Tj, V, Ls = mods
    inp = case["inputs"]()
    ncl = lambda t: t.permute(0, 1, 4, 2, 3).contiguous()  # (B,T,H,W,C) -> (B,T,C,H,W)
    ncl_back = lambda t: t.permute(0, 1, 3, 4, 2).contiguous()

    if name.startswith("morphfc"):
        C = inp["x"].shape[-1]
        ch = case["chunk_of"]("gamma_h")
        m = Fn.Enhanced_MorphFCs_decay(dim=C, chunk_h=ch, chunk_w=ch, qkv_bias=True, non_linear=True, gating=True, symm=True,
                                       symm_act=nn.Tanh, relu_scale=True, channel_mixer="rcab")
        shapes = _load_recipe(m, case)
        n = 3 if name == "morphfc_c144_chunk8" else 1
        return shapes, [m(inp["x"]) for _ in range(n)]
    if name == "rcab_c144":
        m = Fn.RCAB(n_feat=144)
        return _load_recipe(m, case), [m(inp["x"])]
    if name == "mlp_cnn_c144":
        m = Fn.Mlp_cnn(in_features=144, act_layer=nn.GELU, exp_r=2, n_groups=1)
        return _load_recipe(m, case), [m(inp["x"])]
    if name == "mlp_cnn_c112_g4":
        m = Fn.Mlp_cnn(in_features=112, act_layer=nn.GELU, exp_r=2, n_groups=4)
        return _load_recipe(m, case), [m(inp["x"])]
    if name == "tab_c144":
        m = Fn.TAB(embed_dim=144, head=4, chunk_h=8, chunk_w=8, mlp_ratio=2, n_groups=1, qkv_bias=True, drop_path=0.0,
                   if_decay=True, non_linear=True, gating=True, symm=True, symm_act=nn.Tanh, relu_scale=True,
                   ffn="ffn_cnn", mixer_type="mlps", mixer_scaling=1.0, channel_mixer="rcab")
        return _load_recipe(m, case), [m(inp["x"]), m(inp["x"])]
    if name == "updown_down":
        m = L.UpdownkeepSampling(144, 144, mode="down")
        return _load_recipe(m, case), [ncl_back(m(ncl(inp["x"])))]
    if name == "updown_up":
        m = L.UpdownkeepSampling(144, 144, mode="up")
        return _load_recipe(m, case), [ncl_back(m(ncl(inp["x"])))]
    if name == "flow_smoothing":
        return {}, [Fn.Mlp_encoder.flow_smoothing(None, inp["flow"], 4)]
    if name == "ltam_wins":
        m = Tj.LTAM_multi_head(embed_dim=144, stride=4, dim=144, mode="wins", head=4, en_field=False, if_scale=True, twins=[2, 2])
        shapes = _load_recipe(m, case)
        nchw = lambda t: t.permute(0, 3, 1, 2).contiguous()
        nt = lambda t: t.permute(0, 1, 4, 2, 3).contiguous()
        o = m(nchw(inp["q"]), nt(inp["keys"]), nchw(inp["anchor"]), nt(inp["vals"]), None, None, inp["loc"], 2)
        return shapes, [o.permute(0, 2, 3, 1)]
    if name == "trajectory_c32":
        m = Tj.Trajectory_multi_head(embed_dim=32, mode="wins", num_blocks=2, frame_stride=2, traj_win=16, head=4, en_field=False,
                                     head_scale=True, feature_refine=None, r_scaling=0.1, twins=[2, 2], ltam=True)
        shapes = _load_recipe(m, case)
        return shapes, [ncl_back(m(ncl(inp["x"]), inp["ff"], inp["fb"]))]
    if name.startswith("spynet"):
        class Wrap(nn.Module):
            def __init__(self):
                super().__init__()
                self.spynet = V.SPyNet(None)
        m = Wrap()
        shapes = _load_recipe(m, case)
        return shapes, [m.spynet(inp["ref"], inp["supp"])]
    if name == "sr_head":
        class Head(nn.Module):
            def __init__(self):
                super().__init__()
                self.upconv1 = nn.Conv2d(144, 576, 3, 1, 1)
                self.upconv2 = nn.Conv2d(144, 256, 3, 1, 1)
                self.HRconv = nn.Conv2d(64, 64, 3, 1, 1)
                self.conv_last = nn.Conv2d(64, 3, 3, 1, 1)
        # the head is four attributes of VMG, not a class of its own: run VMG.forward's lines 629-632 through a
        # real VMG instance with its trunk bypassed
        cfg = case.get("cfg")
        m = Head()
        shapes = _load_recipe(m, case)
        ps = nn.PixelShuffle(2)
        lr = nn.LeakyReLU(0.1)
        y = inp["y"].permute(0, 3, 1, 2)
        o = lr(ps(m.upconv1(y)))
        o = lr(ps(m.upconv2(o)))
        o = m.conv_last(lr(m.HRconv(o)))
        return shapes, [o.permute(0, 2, 3, 1)]
    if name.startswith("swin_"):
        ws = case["window_of"]("")
        heads = 4 if name == "swin_w2_t5" else 8
        m = S3.DecoderLayer(dim=32, input_resolution=heads, depth=2, num_heads=heads, window_size=list(ws), shift_size=None,
                            mlp_ratio=2, qkv_bias=True, is_train=True, if_unfold=False)
        shapes = _load_recipe(m, case)
        return shapes, [ncl_back(m(ncl(inp["x"])))]
    if name == "loss":
        crit = Ls.CharbonnierLoss(eps=1e-12, if_aux_loss=True, aux_ratio=0.005)
        return {}, [crit(inp["x"], inp["y"]).reshape(1)]
    if name == "lr_schedule":
        # the reference's scheduler itself, stepped over a dummy two-group optimizer (group 0 = the lr-0 SPyNet group)
        import importlib.util
        from . import cases as C
        spec = importlib.util.spec_from_file_location("vmg_ref_lrs", os.path.join(REF, "utils", "lr_scheduler.py"))
        LS = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(LS)
        outs = []
        for cfg in C.LR_SCHEDULES.values():
            ps = [torch.nn.Parameter(torch.zeros(1)) for _ in cfg["base"]]
            opt = torch.optim.AdamW([{"params": [p], "lr": b} for p, b in zip(ps, cfg["base"])], lr=2e-4)
            sch = LS.CosineAnnealingLR_Restart(opt, cfg["T_period"], eta_min=cfg["eta_min"], restarts=cfg["restarts"], weights=cfg["weights"])
            rows, want = [], set(cfg["steps"])
            for t in range(max(cfg["steps"]) + 1):
                if t in want:
                    rows.append([g["lr"] for g in opt.param_groups])
                opt.step()
                sch.step()
            outs.append(torch.tensor(rows, dtype=torch.float64).float() * 1e4)
        return {}, outs
    if name == "lr_update":
        # the reference's own Trainer.update_learning_rate, called as an unbound method on a stub `self` (Trainer.__init__ builds a CUDA model
        # and is not run) that holds a real AdamW with the reference's two pre_training groups and the reference's scheduler
        import types
        import tools.Trainer as TR
        from . import cases as C
        outs = []
        for cfg in C.LR_UPDATES.values():
            ps = [torch.nn.Parameter(torch.zeros(1)) for _ in cfg["base"]]
            opt = torch.optim.AdamW([{"params": [p], "lr": b} for p, b in zip(ps, cfg["base"])], lr=2e-4)
            sch = TR.CosineAnnealingLR_Restart(opt, cfg["T_period"], eta_min=cfg["eta_min"], restarts=cfg["restarts"], weights=cfg["weights"])
            stub = types.SimpleNamespace(optimizer=opt, scheduler=sch, recover_flag=False if cfg["reduced_iter"] is not None else None,
                                         rd_iter=cfg["reduced_iter"], train_configs={"pre_training": True, "pre_lr_ratio": cfg["pre_lr_ratio"]},
                                         config={"network": {"flow_fix": cfg["flow_fix"]}})
            stub._get_init_lr = types.MethodType(TR.Trainer._get_init_lr, stub)
            stub._set_lr = types.MethodType(TR.Trainer._set_lr, stub)
            rows = []
            for it in range(cfg["steps"]):
                opt.step()
                TR.Trainer.update_learning_rate(stub, it, warmup_iter=cfg["warmup_iter"])
                rows.append([g["lr"] for g in opt.param_groups])
            outs.append(torch.tensor(rows, dtype=torch.float64).float() * 1e4)
        return {}, outs
    if name.startswith("infer_"):
        # tools/Tester.py's window loops, called as unbound methods on a stub `self` (Tester.__init__ builds a model from a
        # checkpoint path and is not run); the third-party names its module imports come from oracle/_standins
        import types
        import tools.Tester as TT
        from . import infer_oracle as IO

        def tester(model, **kw):
            return types.SimpleNamespace(model=model, scale=4, **kw)
        x = inp["x"]
        if name == "infer_image":
            return {}, [TT.Tester.test_image(tester(IO.fake_sr_model(), test_spatial=[16, 20], overlapped_spatial_length=6), x),
                        TT.Tester.test_image(tester(IO.fake_sr_model(), test_spatial=[16, 20], overlapped_spatial_length=5), x),
                        TT.Tester.test_image(tester(IO.fake_sr_model(), test_spatial=[16, 20], overlapped_spatial_length=6), x[..., :16, :20])]
        def clips(model, nf, of, sp, osl, fn="test_clips", **extra):
            t = tester(model, test_num_frames=nf, overlapped_num_frames=of, test_spatial=sp, overlapped_spatial_length=osl)
            t.test_image = types.MethodType(TT.Tester.test_image, t)
            return getattr(TT.Tester, fn)(t, x, **extra)
        if name == "infer_clips":
            return {}, [clips(IO.fake_sr_model(), 5, 2, [16, 16], 4), clips(IO.fake_sr_model(), 5, 3, None, None),
                        clips(IO.fake_sr_model(), 4, 0, [16, 16], 4)]
        if name == "infer_clips_max":
            out = clips(IO.fake_sr_model(), 4, 2, None, None, fn="test_clips_max", HR=inp["hr"])
            import numpy as np
            u8 = np.round(np.ascontiguousarray(out.cpu().squeeze().clamp(0, 1).numpy().squeeze().transpose(0, 2, 3, 1)) * 255.0).astype(np.uint8)  # Tester.py:249-250
            return {}, [out, torch.from_numpy(u8.astype(np.float32))]
        if name == "infer_vmg_clips":
            m = _vmg_from_cfg(V, case["cfg"])
            shapes = _load_recipe(m, case)
            return shapes, [clips(m, 3, 1, [64, 64], 8)]
    if name.startswith("vmg_"):
        cfg = case["cfg"]
        m = _vmg_from_cfg(V, cfg)
        shapes = _load_recipe(m, case)
        calls = 2 if name == "vmg_tiny_few" else 1
        return shapes, [m(inp["x"]) for _ in range(calls)]
    raise KeyError(name)


def main(argv):
    from . import cases as C
    torch.manual_seed(0)
    mods = _import_reference()
    os.makedirs(GOLD, exist_ok=True)
    names = argv or list(C.CASES)
    for name in names:
        case = C.CASES[name]
        with torch.no_grad():
            shapes, outs = reference_side(name, case, mods)
        C.save_fixture(os.path.join(GOLD, f"{name}.npz"), shapes, outs)
        print(f"{name}: {len(shapes)} tensors, outputs {[tuple(o.shape) for o in outs]}")


if __name__ == "__main__":
    main(sys.argv[1:])
