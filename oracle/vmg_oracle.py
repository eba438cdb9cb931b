"""CPU oracle for the VMG per-frame forward hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch fp32 restatement of the reference's algorithm (EasyVision-Ton/VMG,
reference @ 2024_10_08).  It exists so that the HIP product path in ``vmg_amd/`` can be checked against
an independent implementation on machines where the reference itself cannot travel (the GPU box).

Rules (see DESIGN.md "Oracle"):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it;
  * it is pinned against the reference itself: ``oracle/gen_golden.py`` imports the unmodified reference
    ``models/*.py`` in the build container and writes small fixtures under ``tests/golden/`` which
    ``tests/test_oracle_golden.py`` replays through this file (no reference needed at test time);
  * the reference publishes no golden vectors or tests of its own.

Design: purely functional.  Every function takes the reference-format ``state_dict`` (``sd``) plus a key
prefix, so the very same recipe weights drive the reference, this oracle and the HIP module.  Features
are kept channels-last ``(N, H, W, C)`` between ops; convolutions go through ``F.conv2d`` on an NCHW view.

Statefulness (SURVEY trap T1): the reference multiplies ``mlp_h[0].weight`` / ``mlp_w[0].weight`` by the
``gamma_h`` / ``gamma_w`` buffers in place on EVERY forward call (models/function.py:766-768,779-781).
``morphfc_decay`` below does the same to the tensors inside ``sd`` so that call #k of the oracle equals
call #k of the reference.  Pass ``mutate=False`` to evaluate with explicit call index instead.

Citations are ``file:line`` relative to /root/reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from fractions import Fraction
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------------------------------
# configuration (mirrors VMG.__init__ keyword arguments, models/vmg.py:177-210)
# --------------------------------------------------------------------------------------------------
@dataclass
class VMGConfig:
    embed_dim: Sequence[int] = (144, 144, 144)
    depths: Sequence[int] = (4, 4, 4)
    num_heads: Sequence[int] = (4, 8, 4)
    num_frames: int = 7
    window_sizes: Sequence[Sequence[int]] = ((2, 8, 8), (4, 8, 8), (2, 8, 8))
    mlp_ratio: float = 2
    n_groups: int = 1
    image_size: Sequence[int] = (64, 64)
    is_train: bool = False
    traj_win: Sequence[Optional[int]] = (16, None)
    traj_keyframes_n: Sequence[Optional[int]] = (3, None)
    traj_heads: Sequence[Optional[int]] = (4, None)
    temporal_type: Sequence[Optional[bool]] = (False, None)
    temporal_empty: bool = True
    traj_res_n: Sequence[int] = (15, 0, 15)
    spatial_type: Sequence[bool] = (False, False)
    mdsc: bool = False
    if_concat: bool = False
    flow_smooth: bool = True
    smooth_region_range: int = 4
    r_scaling: float = 0.1
    chunk_ratios: Sequence = ("1/8", "1/4")
    twins: Sequence[int] = (2, 2)
    traj_scale: bool = True
    m_scaling: float = 1.0
    if_local_fuse: bool = True
    channel_mixer: str = "rcab"
    ffn_type: str = "ffn_cnn"
    back_RBs: int = 0

    # derived (models/vmg.py:213-216, 243-245)
    @property
    def num_layers(self):
        return len(self.depths)

    @property
    def num_enc_layers(self):
        return self.num_layers // 2 + 1

    @property
    def num_dec_layers(self):
        return self.num_layers // 2

    @property
    def scale(self):
        return 2 ** (self.num_enc_layers - 1)

    @property
    def chunk_h(self):
        return [int(self.image_size[0] * float(Fraction(r))) for r in self.chunk_ratios]

    @property
    def chunk_w(self):
        return [int(self.image_size[1] * float(Fraction(r))) for r in self.chunk_ratios]


# --------------------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------------------
def conv_nhwc(x: Tensor, w: Tensor, b: Optional[Tensor], pad: int, groups: int = 1) -> Tensor:
    """Conv2d on channels-last features: (N,H,W,Cin) -> (N,H,W,Cout).  Weight is OIHW as in the checkpoint."""
    y = F.conv2d(x.permute(0, 3, 1, 2), w, b, stride=1, padding=pad, groups=groups)
    return y.permute(0, 2, 3, 1)


def flow_warp(x: Tensor, flow: Tensor, mode: str = "bilinear", padding: str = "zeros") -> Tensor:
    """x (n,c,h,w) sampled at pixel grid + flow (n,h,w,2), align_corners=True.

    Restates the three identical copies at models/vmg.py:640-685, models/trajectory.py:71-116,
    models/function.py:1546-1582 (normalisation 2x/max(w-1,1)-1 then F.grid_sample).
    """
    n, c, h, w = x.shape
    if (h, w) != tuple(flow.shape[1:3]):
        raise ValueError(f"The spatial sizes of input ({x.size()[-2:]}) and flow ({flow.size()[1:3]}) are not the same.")
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    base = torch.stack((xs, ys), 2).to(x.dtype)
    g = base + flow
    gx = 2.0 * g[..., 0] / max(w - 1, 1) - 1.0
    gy = 2.0 * g[..., 1] / max(h - 1, 1) - 1.0
    return F.grid_sample(x, torch.stack((gx, gy), 3).to(x.dtype), mode=mode, padding_mode=padding, align_corners=True)


# --------------------------------------------------------------------------------------------------
# SPyNet (models/vmg.py:18-173)
# --------------------------------------------------------------------------------------------------
def spynet_basic(sd: SD, p: str, x: Tensor) -> Tensor:
    """5x conv7x7 (+ReLU on the first four).  models/vmg.py:126-173; ConvModule = Conv2d+bias+ReLU."""
    for j in range(5):
        x = F.conv2d(x, sd[f"{p}basic_module.{j}.conv.weight"], sd[f"{p}basic_module.{j}.conv.bias"], padding=3)
        if j < 4:
            x = F.relu(x)
    return x


def spynet_compute_flow(sd: SD, p: str, ref: Tensor, supp: Tensor) -> Tensor:
    """Coarse-to-fine residual flow.  models/vmg.py:39-85."""
    n, _, h, w = ref.shape
    mean, std = sd[f"{p}mean"], sd[f"{p}std"]
    refs = [(ref - mean) / std]
    supps = [(supp - mean) / std]
    for _ in range(5):
        refs.append(F.avg_pool2d(refs[-1], 2, 2, count_include_pad=False))
        supps.append(F.avg_pool2d(supps[-1], 2, 2, count_include_pad=False))
    refs, supps = refs[::-1], supps[::-1]
    flow = ref.new_zeros(n, 2, h // 32, w // 32)
    for lvl in range(6):
        if lvl == 0:
            up = flow
        else:
            up = F.interpolate(flow, scale_factor=2, mode="bilinear", align_corners=True) * 2.0
        warped = flow_warp(supps[lvl], up.permute(0, 2, 3, 1), padding="border")
        flow = up + spynet_basic(sd, f"{p}basic_module.{lvl}.", torch.cat([refs[lvl], warped, up], 1))
    return flow


def spynet_forward(sd: SD, p: str, ref: Tensor, supp: Tensor) -> Tensor:
    """Flow ref->supp at input resolution.  models/vmg.py:87-123."""
    h, w = ref.shape[2:4]
    w_up = w if w % 32 == 0 else 32 * (w // 32 + 1)
    h_up = h if h % 32 == 0 else 32 * (h // 32 + 1)
    ref_u = F.interpolate(ref, size=(h_up, w_up), mode="bilinear", align_corners=False)
    supp_u = F.interpolate(supp, size=(h_up, w_up), mode="bilinear", align_corners=False)
    flow = F.interpolate(spynet_compute_flow(sd, p, ref_u, supp_u), size=(h, w), mode="bilinear", align_corners=False)
    sx, sy = float(w) / float(w_up), float(h) / float(h_up)
    return torch.stack((flow[:, 0] * sx, flow[:, 1] * sy), 1)


def frames_mirror(x: Tensor) -> bool:
    """models/vmg.py:426-432."""
    if x.size(1) % 2 != 0:
        return False
    a, b = torch.chunk(x, 2, dim=1)
    return bool(torch.linalg.norm(a - b.flip(1)) == 0)


def compute_flows(sd: SD, cfg: VMGConfig, x: Tensor):
    """Per encoder scale: forward and backward flows (B,T-1,2,h,w).  models/vmg.py:435-464."""
    B, T, C, H, W = x.shape
    mirror = frames_mirror(x)
    fwd, bwd = [], []
    for i in range(cfg.num_enc_layers):
        h, w = H // (2 ** i), W // (2 ** i)
        xi = F.adaptive_avg_pool2d(x.reshape(B * T, C, H, W), (h, w)).reshape(B, T, C, h, w)
        a = xi[:, :-1].reshape(-1, C, h, w)
        b = xi[:, 1:].reshape(-1, C, h, w)
        ff = spynet_forward(sd, "spynet.", b, a).view(B, T - 1, 2, h, w)
        if mirror:
            fb = ff.flip(1)
        else:
            fb = spynet_forward(sd, "spynet.", a, b).view(B, T - 1, 2, h, w)
        fwd.append(ff)
        bwd.append(fb)
    return fwd, bwd


# --------------------------------------------------------------------------------------------------
# TAB pieces (models/function.py)
# --------------------------------------------------------------------------------------------------
def decay_gamma(chunk: int, ch_total: int) -> Tensor:
    """The (Ch,Ch) retention matrix registered as gamma_h / gamma_w.

    Closed form of Enhanced_MorphFCs_decay.form_decay (models/function.py:651-652, 684-732):
    with S = Ch/chunk, d_i = exp(log(1 - 2^-(5 + chunk-1-i))), Gamma[r, c] = mean_i d_i^(|r//S - c//S| + 1),
    powers formed by repeated multiplication as the reference's cumulative product does.
    """
    S = ch_total // chunk
    d = torch.log(1 - 2 ** (-5 - torch.arange(chunk - 1, -1, -1, dtype=torch.float))).exp()  # (chunk,)
    pw = [d.clone()]
    for _ in range(1, chunk):
        pw.append(pw[-1] * d)
    pw = torch.stack(pw, 1)  # (i, k) = d_i^(k+1)
    idx = torch.arange(chunk)
    dist = (idx[:, None] - idx[None, :]).abs()  # (j, s)
    blk = pw[:, dist]  # (i, j, s)
    full = blk[:, :, None, :, None].expand(chunk, chunk, S, chunk, S).reshape(chunk, chunk * S, chunk * S)
    return torch.mean(full, 0)


def morph_tokens(x: Tensor, axis: str, chunk: int, Cp: int) -> Tensor:
    """Token layout of the H- or W-branch (models/function.py:763-764, 776-777).

    x (B,T,H,W,C).  Pads C to Cp and the mixed axis to a multiple of `chunk`; groups `chunk` consecutive
    positions along the axis; splits channels into `chunk` chunks of S = Cp/chunk; token (group, k) has
    features f = p*S + s  <-  x[position p of the group, channel k*S + s].
    Returns (B,T,G,chunk,Cp) with G = number of groups (for H: W*Hp/chunk, W-major).
    """
    B, T, H, W, C = x.shape
    S = Cp // chunk
    if axis == "h":
        Hp = int(math.ceil(H / chunk)) * chunk
        xp = F.pad(x, (0, Cp - C, 0, 0, 0, Hp - H)).transpose(2, 3)  # (B,T,W,Hp,Cp)
        L = W * Hp
    else:
        Wp = int(math.ceil(W / chunk)) * chunk
        xp = F.pad(x, (0, Cp - C, 0, Wp - W))  # (B,T,H,Wp,Cp)
        L = H * Wp
    t = xp.reshape(B, T, L // chunk, chunk, chunk, S)  # [group, p, k, s]
    return t.permute(0, 1, 2, 4, 3, 5).reshape(B, T, L // chunk, chunk, chunk * S)


def morph_untokens(t: Tensor, axis: str, chunk: int, Cp: int, H: int, W: int, C: int) -> Tensor:
    """Inverse of morph_tokens followed by the crop (models/function.py:772, 785)."""
    B, T = t.shape[:2]
    S = Cp // chunk
    G = t.shape[2]
    u = t.reshape(B, T, G, chunk, chunk, S).permute(0, 1, 2, 4, 3, 5)  # [group, p, k, s]
    if axis == "h":
        Hp = int(math.ceil(H / chunk)) * chunk
        return u.reshape(B, T, W, Hp, Cp).transpose(2, 3)[..., 0:H, :, :C]
    Wp = int(math.ceil(W / chunk)) * chunk
    return u.reshape(B, T, H, Wp, Cp)[..., 0:W, :C]


def rcab(sd: SD, p: str, x: Tensor) -> Tensor:
    """RCAB + CALayer on (B,T,H,W,C).  models/function.py:561-583, 542-558 (reduction 8)."""
    B, T, H, W, C = x.shape
    f = x.reshape(B * T, H, W, C)
    r = F.relu(conv_nhwc(f, sd[f"{p}body.0.weight"], sd[f"{p}body.0.bias"], 1))
    r = conv_nhwc(r, sd[f"{p}body.2.weight"], sd[f"{p}body.2.bias"], 1)
    g = r.mean((1, 2))  # GAP -> (N,C)
    g = F.relu(F.linear(g, sd[f"{p}body.3.conv_du.0.weight"].flatten(1), sd[f"{p}body.3.conv_du.0.bias"]))
    g = torch.sigmoid(F.linear(g, sd[f"{p}body.3.conv_du.2.weight"].flatten(1), sd[f"{p}body.3.conv_du.2.bias"]))
    r = r * g[:, None, None, :] + f
    return r.reshape(B, T, H, W, C)


def morphfc_decay(sd: SD, p: str, x: Tensor, chunk_h: int, chunk_w: int, mutate: bool = True,
                  call_index: int = 1) -> Tensor:
    """Enhanced_MorphFCs_decay.forward (models/function.py:743-805), config: non_linear, gating+symm tanh,
    relu_scale; channel_mixer 'rcab' or 'vanilla' (decided by the keys present in ``sd``, as the module tree does).

    mutate=True reproduces the in-place weight decay (T1).  mutate=False leaves ``sd`` untouched and uses
    W * Gamma**call_index (elementwise power by repeated multiplication).
    """
    B, T, H, W, C = x.shape
    Ch = int(math.ceil(C / chunk_h)) * chunk_h
    Cw = int(math.ceil(C / chunk_w)) * chunk_w

    def decayed(name, gname):
        if mutate:
            sd[name].mul_(sd[gname])
            return sd[name]
        wv = sd[name]
        for _ in range(call_index):
            wv = wv * sd[gname]
        return wv

    wh = decayed(f"{p}mlp_h.0.weight", f"{p}gamma_h")
    th = morph_tokens(x, "h", chunk_h, Ch)
    th = F.relu(F.linear(th, wh, sd[f"{p}mlp_h.0.bias"])) / Ch
    h = morph_untokens(th, "h", chunk_h, Ch, H, W, C)

    ww = decayed(f"{p}mlp_w.0.weight", f"{p}gamma_w")
    tw = morph_tokens(x, "w", chunk_w, Cw)
    tw = F.relu(F.linear(tw, ww, sd[f"{p}mlp_w.0.bias"])) / Cw
    w = morph_untokens(tw, "w", chunk_w, Cw, H, W, C)

    if f"{p}mlp_c.0.weight" in sd:  # channel_mixer 'vanilla' + non_linear: Sequential(Linear, ReLU) (models/function.py:640-644)
        c = F.relu(F.linear(x, sd[f"{p}mlp_c.0.weight"], sd[f"{p}mlp_c.0.bias"])) / C
    else:  # channel_mixer 'rcab' (models/function.py:645-646)
        c = rcab(sd, f"{p}mlp_c.", x) / C

    a = (h + w + c).mean((1, 2, 3))  # (B,C)
    a = F.linear(a, sd[f"{p}reweight.fc1.weight"], sd[f"{p}reweight.fc1.bias"])
    a = F.linear(F.gelu(a), sd[f"{p}reweight.fc2.weight"], sd[f"{p}reweight.fc2.bias"])
    a = a.reshape(B, C, 3).softmax(-1)  # weights over (h, w, c) per (b, channel)
    a = a[:, None, None, None]  # (B,1,1,1,C,3)
    y = h * a[..., 0] + w * a[..., 1] + c * a[..., 2]
    y = F.linear(y, sd[f"{p}proj.weight"], sd[f"{p}proj.bias"])
    return (x + y) * torch.tanh(y)


def mlp_cnn(sd: SD, p: str, x: Tensor, n_groups: int) -> Tensor:
    """Mlp_cnn.forward: conv3x3(C->rC, groups)+GELU(erf), Linear(rC->C).  models/function.py:67-79."""
    B, T, H, W, C = x.shape
    y = conv_nhwc(x.reshape(B * T, H, W, C), sd[f"{p}fc1.weight"], sd[f"{p}fc1.bias"], 1, n_groups)
    y = F.linear(F.gelu(y), sd[f"{p}fc2.weight"], sd[f"{p}fc2.bias"])
    return y.reshape(B, T, H, W, C)


def mlp_vanilla(sd: SD, p: str, x: Tensor) -> Tensor:
    """Mlp.forward (fc1-GELU-fc2).  models/function.py:41-47."""
    return F.linear(F.gelu(F.linear(x, sd[f"{p}fc1.weight"], sd[f"{p}fc1.bias"])), sd[f"{p}fc2.weight"], sd[f"{p}fc2.bias"])


def tab(sd: SD, p: str, x: Tensor, cfg: VMGConfig, chunk_h: int, chunk_w: int, mutate=True, call_index=1) -> Tensor:
    """TAB.forward with DropPath = identity (eval / p = 0).  models/function.py:1212-1217."""
    C = x.shape[-1]
    s = cfg.m_scaling
    n2 = F.layer_norm(x, (C,), sd[f"{p}norm2.weight"], sd[f"{p}norm2.bias"], 1e-5)
    x = x + morphfc_decay(sd, f"{p}spatial_mixing.", n2, chunk_h, chunk_w, mutate, call_index) * s
    n3 = F.layer_norm(x, (C,), sd[f"{p}norm3.weight"], sd[f"{p}norm3.bias"], 1e-5)
    if cfg.ffn_type == "ffn_cnn":
        x = x + mlp_cnn(sd, f"{p}channel_mixing.", n3, cfg.n_groups) * s
    else:
        x = x + mlp_vanilla(sd, f"{p}channel_mixing.", n3) * s
    return x


def flow_smoothing(flow: Tensor, r: int) -> Tensor:
    """Reflect-pad to a multiple of r, r x r mean, nearest x r, crop.  models/function.py:1466-1478."""
    B, T, C, H, W = flow.shape
    f = flow.reshape(-1, C, H, W)
    hf, wf = int(math.ceil(H / r)) * r, int(math.ceil(W / r)) * r
    f = F.pad(f, (0, wf - W, 0, hf - H), mode="reflect")
    f = F.adaptive_avg_pool2d(f, (hf // r, wf // r))
    f = F.interpolate(f, scale_factor=r, mode="nearest")[..., :H, :W]
    return f.reshape(B, T, C, H, W)


# --------------------------------------------------------------------------------------------------
# trajectory attention + recurrent residual chains (models/trajectory.py)
# --------------------------------------------------------------------------------------------------
def resblocks(sd: SD, p: str, x: Tensor, num_blocks: int, r_scaling: float) -> Tensor:
    """ResidualBlocksWithInputConv on NHWC features.  models/trajectory.py:16-52, 165-221."""
    x = F.leaky_relu(conv_nhwc(x, sd[f"{p}main.0.weight"], sd[f"{p}main.0.bias"], 1), 0.1)
    for k in range(num_blocks):
        q = f"{p}main.2.{k}."
        y = F.relu(conv_nhwc(x, sd[f"{q}conv1.weight"], sd[f"{q}conv1.bias"], 1))
        y = conv_nhwc(y, sd[f"{q}conv2.weight"], sd[f"{q}conv2.bias"], 1)
        x = x + y * r_scaling
    return x


def ltam_rpe(sd: SD, p: str, t: int, qn: int) -> Tensor:
    """decay^age * RPE, (head, qn, t*qn).  models/trajectory.py:534-547: key-frame j of t (0 = oldest) gets
    decay_v^(t-j) formed by repeated multiplication."""
    dv = sd[f"{p}decay_v"]
    pw = [dv.clone()]
    for _ in range(1, t):
        pw.append(pw[-1] * dv)
    pw = torch.stack(pw[::-1], 1)  # (head, t): index j -> dv^(t-j)
    rpe = sd[f"{p}relative_pos_encoding"]  # (head, qn, qn)
    return (pw[:, None, :, None] * rpe[:, :, None, :]).reshape(rpe.shape[0], qn, t * qn)


def ltam_wins(sd: SD, p: str, q: Tensor, keys: Tensor, anchor: Tensor, vals: Tensor, loc: Tensor,
              heads: int, twins: Sequence[int], scale_on: bool = True) -> Tensor:
    """LTAM_multi_head.forward_wins (models/trajectory.py:672-795), en_field False, ia False.

    q, anchor (n,h,w,c); keys, vals (n,t,h,w,c) = key-frame inputs / key-frame states; loc (n,2t,h,w) tracked
    pixel coordinates (x then y per key-frame).  Returns (n,h,w,c).
    """
    n, h, w, c = anchor.shape
    t = vals.shape[1]
    wh, ww = twins
    d = c // heads
    scale = d ** -0.5 if scale_on else 1.0
    g = loc.reshape(n, t, 2, h, w).permute(0, 1, 3, 4, 2)
    gx = 2.0 * g[..., 0] / max(w - 1, 1) - 1.0
    gy = 2.0 * g[..., 1] / max(h - 1, 1) - 1.0
    grid = torch.stack((gx, gy), 4).reshape(n * t, h, w, 2)

    def gather(src):  # nearest, zeros, align_corners=True
        o = F.grid_sample(src.reshape(n * t, h, w, c).permute(0, 3, 1, 2), grid.to(src.dtype), mode="nearest",
                          padding_mode="zeros", align_corners=True)
        return o.permute(0, 2, 3, 1).reshape(n, t, h, w, c)

    v = gather(vals)
    k = F.normalize(gather(keys), dim=-1)
    qn = F.normalize(q, dim=-1)

    def windows(z):  # (n, [t,] h, w, c) -> (n, nwin, heads, [t*]wh*ww, d)
        if z.dim() == 4:
            z = z[:, None]
        tt = z.shape[1]
        z = z.reshape(n, tt, h // wh, wh, w // ww, ww, heads, d)
        return z.permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(n, (h // wh) * (w // ww), heads, tt * wh * ww, d)

    qw, kw, vw = windows(qn), windows(k), windows(v)
    logits = torch.matmul(qw * scale, kw.transpose(-1, -2))  # (n, nwin, heads, qn, t*qn)
    logits = logits + ltam_rpe(sd, p, t, wh * ww)[None, None]
    out = torch.matmul(logits.softmax(-1), vw)  # (n, nwin, heads, qn, d)
    out = out.reshape(n, h // wh, w // ww, heads, wh, ww, d).permute(0, 1, 4, 2, 5, 3, 6).reshape(n, h, w, c)
    return F.linear(out, sd[f"{p}proj.weight"], sd[f"{p}proj.bias"]) + anchor


def trajectory(sd: SD, p: str, x: Tensor, flows_f: Tensor, flows_b: Tensor, cfg: VMGConfig, stage: int,
               num_blocks: int) -> Tensor:
    """Trajectory_multi_head.forward (models/trajectory.py:300-490); x (B,T,H,W,C), flows (B,T-1,2,H,W)."""
    n, t, h, w, c = x.shape
    stride = cfg.traj_keyframes_n[stage]
    heads = cfg.traj_heads[stage]
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    ident = torch.stack([xs, ys], 0).to(x.dtype)[None].expand(n, -1, -1, -1)

    def sweep(order, flow_of, key_idx):
        feat = x.new_zeros(n, h, w, c)
        loc = ident
        k_in, k_state, outs = [], [], {}
        for step, i in enumerate(order):
            cur = x[:, i]
            if step > 0:
                fl = flow_of(i).permute(0, 2, 3, 1)
                feat = flow_warp(feat.permute(0, 3, 1, 2), fl, padding="border").permute(0, 2, 3, 1)
                loc = flow_warp(loc, fl, mode="nearest", padding="border")
                feat = ltam_wins(sd, f"{p}LTAM.", cur, torch.stack(k_in, 1), feat, torch.stack(k_state, 1), loc,
                                 heads, cfg.twins, cfg.traj_scale)
                if i in key_idx:
                    loc = torch.cat([loc, ident], 1)
            feat = resblocks(sd, f"{p}resblocks.", torch.cat([cur, feat], -1), num_blocks, cfg.r_scaling)
            if i in key_idx:
                k_state.append(feat)
                k_in.append(cur)
            outs[i] = feat
        return outs

    back = sweep(list(range(t - 1, -1, -1)), lambda i: flows_b[:, i], list(range(t - 1, -1, -stride)))
    fwd = sweep(list(range(t)), lambda i: flows_f[:, i - 1], list(range(0, t, stride)))
    wf = sd[f"{p}fusion.weight"].flatten(1)
    outs = [F.leaky_relu(F.linear(torch.cat([back[i], x[:, i], fwd[i]], -1), wf, sd[f"{p}fusion.bias"]), 0.1)
            for i in range(t)]
    return torch.stack(outs, 1)


# --------------------------------------------------------------------------------------------------
# 3-D shifted-window attention (models/swin_3d.py), reachable with temporal_empty=False
# --------------------------------------------------------------------------------------------------
def get_window_size(x_size, window_size, shift_size):
    """models/swin_3d.py:88-101."""
    ws, ss = list(window_size), list(shift_size)
    for i in range(3):
        if x_size[i] <= window_size[i]:
            ws[i] = x_size[i]
            ss[i] = 0
    return tuple(ws), tuple(ss)


def window_partition(x: Tensor, ws) -> Tensor:
    """(B,D,H,W,C) -> (B*nW, wt*wh*ww, C).  models/swin_3d.py:55-68."""
    B, D, H, W, C = x.shape
    x = x.reshape(B, D // ws[0], ws[0], H // ws[1], ws[1], W // ws[2], ws[2], C)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(-1, ws[0] * ws[1] * ws[2], C)


def window_reverse(win: Tensor, ws, B, D, H, W) -> Tensor:
    """models/swin_3d.py:71-85."""
    x = win.reshape(B, D // ws[0], H // ws[1], W // ws[2], ws[0], ws[1], ws[2], -1)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, D, H, W, -1)


def shift_mask(D, H, W, ws, ss) -> Tensor:
    """(nW, N, N) mask of 0 / -100.  models/swin_3d.py:104-118."""
    img = torch.zeros(1, D, H, W, 1)
    cnt = 0
    for d in (slice(-ws[0]), slice(-ws[0], -ss[0]), slice(-ss[0], None)):
        for h in (slice(-ws[1]), slice(-ws[1], -ss[1]), slice(-ss[1], None)):
            for w in (slice(-ws[2]), slice(-ws[2], -ss[2]), slice(-ss[2], None)):
                img[:, d, h, w, :] = cnt
                cnt += 1
    mw = window_partition(img, ws).squeeze(-1)
    diff = mw[:, None, :] - mw[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def relative_position_index(ws) -> Tensor:
    """models/swin_3d.py:309-323."""
    coords = torch.stack(torch.meshgrid(torch.arange(ws[0]), torch.arange(ws[1]), torch.arange(ws[2]), indexing="ij"))
    cf = coords.flatten(1)
    rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws[0] - 1
    rel[:, :, 1] += ws[1] - 1
    rel[:, :, 2] += ws[2] - 1
    rel[:, :, 0] *= (2 * ws[1] - 1) * (2 * ws[2] - 1)
    rel[:, :, 1] *= 2 * ws[2] - 1
    return rel.sum(-1)


def rwindow_attention(sd: SD, p: str, xw: Tensor, mask: Optional[Tensor], heads: int, ws_cfg) -> Tensor:
    """rWindowAttention.forward, only_one=False (models/swin_3d.py:167-252): the queries of each time slice
    attend to the tokens of the OTHER slices of the window.  ws_cfg is the constructor window size (it fixes
    the slice length `interval` and the bias table indexing)."""
    B_, N, C = xw.shape
    d = C // heads
    scale = d ** -0.5
    interval = ws_cfg[1] * ws_cfg[2]
    total = ws_cfg[0] * ws_cfg[1] * ws_cfg[2]
    q = F.linear(xw, sd[f"{p}q.weight"], sd.get(f"{p}q.bias")).reshape(B_, N, heads, d).permute(0, 2, 1, 3)
    kv = F.linear(xw, sd[f"{p}kv.weight"], sd.get(f"{p}kv.bias")).reshape(B_, N, 2, heads, d).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    table, index = sd[f"{p}relative_position_bias_table"], sd[f"{p}relative_position_index"]
    seq = list(range(total))
    outs = []
    for i in range(len(range(0, total, interval))):
        lo = i * interval
        hi = total if i == len(range(0, total, interval)) - 1 else (i + 1) * interval
        other = [s for s in seq if s < lo or s >= hi]
        attn = (q[:, :, lo:hi] * scale) @ k[:, :, other].transpose(-2, -1)
        bias = table[index[lo:hi][:, other].reshape(-1)].reshape(hi - lo, len(other), heads).permute(2, 0, 1)
        attn = attn + bias[None]
        if mask is not None:
            nW = mask.shape[0]
            attn = attn.reshape(B_ // nW, nW, heads, hi - lo, len(other)) + mask[:, lo:hi][:, :, other][None, :, None]
            attn = attn.reshape(-1, heads, hi - lo, len(other))
        outs.append((attn.softmax(-1) @ v[:, :, other]).transpose(1, 2).reshape(B_, hi - lo, C))
    return F.linear(torch.cat(outs, 1), sd[f"{p}proj.weight"], sd[f"{p}proj.bias"])


def swin_block(sd: SD, p: str, x: Tensor, mask_full: Tensor, heads: int, ws_cfg, shift_cfg) -> Tensor:
    """EncoderBlockOnOnetoken.forward, if_unfold False, drop_path 0 (models/swin_3d.py:772-855)."""
    B, D, H, W, C = x.shape
    ws, ss = get_window_size((D, H, W), ws_cfg, shift_cfg)
    y = F.layer_norm(x, (C,), sd[f"{p}norm1.weight"], sd[f"{p}norm1.bias"], 1e-5)
    pd = (ws[0] - D % ws[0]) % ws[0]
    pb = (ws[1] - H % ws[1]) % ws[1]
    pr = (ws[2] - W % ws[2]) % ws[2]
    y = F.pad(y, (0, 0, 0, pr, 0, pb, 0, pd))
    _, Dp, Hp, Wp, _ = y.shape
    shifted = any(s > 0 for s in ss)
    if shifted:
        y = torch.roll(y, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3))
    aw = rwindow_attention(sd, f"{p}attn.", window_partition(y, ws), mask_full if shifted else None, heads, ws_cfg)
    y = window_reverse(aw.reshape(-1, *ws, C), ws, B, Dp, Hp, Wp)
    if shifted:
        y = torch.roll(y, shifts=ss, dims=(1, 2, 3))
    x = x + y[:, :D, :H, :W]
    z = F.layer_norm(x, (C,), sd[f"{p}norm2.weight"], sd[f"{p}norm2.bias"], 1e-5)
    z = F.linear(F.gelu(F.linear(z, sd[f"{p}mlp.fc1.weight"], sd[f"{p}mlp.fc1.bias"])), sd[f"{p}mlp.fc2.weight"],
                 sd[f"{p}mlp.fc2.bias"])
    return x + z


def swin_decoder_layer(sd: SD, p: str, x: Tensor, heads: int, ws_cfg, depth: int = 2) -> Tensor:
    """swin_3d.DecoderLayer.forward (models/swin_3d.py:1141-1202) on (B,D,H,W,C) features.

    Temporal padding: when D is not a multiple of wt the last full-window frames are repeated in reverse
    (rearrange_shape) and removed again afterwards.
    """
    ws_cfg = list(ws_cfg)
    shift_cfg = [i // 2 for i in ws_cfg]
    B, D, H, W, C = x.shape
    seq_back = None
    wt = ws_cfg[0]
    if D % wt != 0:
        delta_t = int(math.ceil(D / wt)) * wt - D
        delta = list(range(-1, -(delta_t + 1), -1))
        start = list(range(0, (D // wt) * wt))
        end = list(range((D // wt) * wt, D))
        new_seq = start + delta + end
        seq_back = start + list(range(-1, -(len(end) + 1), -1))[::-1]
        rep = [start[i] for i in delta]
        x = torch.cat([x, x[:, rep]], 1)[:, new_seq]
        D = x.shape[1]
    ws, ss = get_window_size((D, H, W), ws_cfg, shift_cfg)
    Dp = int(math.ceil(D / ws[0])) * ws[0]
    Hp = int(math.ceil(H / ws[1])) * ws[1]
    Wp = int(math.ceil(W / ws[2])) * ws[2]
    mask = shift_mask(Dp, Hp, Wp, ws, ss)
    for i in range(depth):
        x = swin_block(sd, f"{p}blocks.{i}.", x, mask, heads, ws_cfg, [0, 0, 0] if i % 2 == 0 else shift_cfg)
    if seq_back is not None:
        x = x[:, seq_back]
    return x


# --------------------------------------------------------------------------------------------------
# stage container, sampling, head, whole model
# --------------------------------------------------------------------------------------------------
def mlp_encoder(sd: SD, p: str, x: Tensor, flow_f: Optional[Tensor], flow_b: Optional[Tensor], cfg: VMGConfig,
                depth: int, chunk_h: int, chunk_w: int, aligned, traj_stage: int, traj_blocks: int, heads: int,
                window_size, mutate=True, call_index=1) -> Tensor:
    """Mlp_encoder.forward on channels-last (B,T,H,W,C) in and out.  models/function.py:1480-1543."""
    B, T, H, W, C = x.shape
    short = x
    if flow_f is not None and cfg.flow_smooth:
        flow_b = flow_smoothing(flow_b, cfg.smooth_region_range)
        flow_f = flow_smoothing(flow_f, cfg.smooth_region_range)
    for j in range(depth):
        x = tab(sd, f"{p}mlp_blocks.{j}.", x, cfg, chunk_h, chunk_w, mutate, call_index)
    if cfg.if_local_fuse:
        y = conv_nhwc(x.reshape(B * T, H, W, C), sd[f"{p}local_cnn.weight"], sd[f"{p}local_cnn.bias"], 1)
        x = short + y.reshape(B, T, H, W, C)
    if aligned is None:
        if not cfg.temporal_empty:
            x = swin_decoder_layer(sd, f"{p}traj_mixing.", x, heads, window_size)
    elif aligned is False:
        x = trajectory(sd, f"{p}traj_mixing.", x, flow_f, flow_b, cfg, traj_stage, traj_blocks)
    else:
        raise NotImplementedError("aligned=True (DCN) is not configured by any shipped config")
    return x


def updown(sd: SD, p: str, x: Tensor, mode: str) -> Tensor:
    """UpdownkeepSampling on channels-last features; channel order (neiw neih c).  models/layers.py:777-798."""
    B, T, H, W, C = x.shape
    if mode == "down":
        y = x.reshape(B, T, H // 2, 2, W // 2, 2, C).permute(0, 1, 2, 4, 5, 3, 6).reshape(B, T, H // 2, W // 2, 4 * C)
    else:
        c4 = C // 4
        y = x.reshape(B, T, H, W, 2, 2, c4).permute(0, 1, 2, 5, 3, 4, 6).reshape(B, T, 2 * H, 2 * W, c4)
    y = F.layer_norm(y, (y.shape[-1],), sd[f"{p}norm.weight"], sd[f"{p}norm.bias"], 1e-5)
    return F.linear(y, sd[f"{p}linear.weight"], sd[f"{p}linear.bias"])


def pixel_shuffle_nhwc(x: Tensor, r: int = 2) -> Tensor:
    """torch PixelShuffle order (channel = c*r*r + i*r + j) on NHWC.  models/vmg.py:380, 629-630."""
    N, H, W, C = x.shape
    c = C // (r * r)
    return x.reshape(N, H, W, c, r, r).permute(0, 1, 4, 2, 5, 3).reshape(N, H * r, W * r, c)


def sr_head(sd: SD, y: Tensor) -> Tensor:
    """upconv1/PS/lrelu, upconv2/PS/lrelu, HRconv/lrelu, conv_last on (N,H,W,C) -> (N,4H,4W,3).  vmg.py:629-632."""
    o = F.leaky_relu(pixel_shuffle_nhwc(conv_nhwc(y, sd["upconv1.weight"], sd["upconv1.bias"], 1)), 0.1)
    o = F.leaky_relu(pixel_shuffle_nhwc(conv_nhwc(o, sd["upconv2.weight"], sd["upconv2.bias"], 1)), 0.1)
    o = F.leaky_relu(conv_nhwc(o, sd["HRconv.weight"], sd["HRconv.bias"], 1), 0.1)
    return conv_nhwc(o, sd["conv_last.weight"], sd["conv_last.bias"], 1)


def mdsc_skip(sd: SD, p: str, x: Tensor) -> Tensor:
    """adaptive_max_pool /4 -> conv1x1 -> GroupNorm(1) -> ReLU.  models/vmg.py:389-400, 519, 525."""
    B, T, H, W, C = x.shape
    f = F.adaptive_max_pool2d(x.reshape(B * T, H, W, C).permute(0, 3, 1, 2), (H // 4, W // 4))
    f = F.conv2d(f, sd[f"{p}0.weight"], sd[f"{p}0.bias"])
    f = F.relu(F.group_norm(f, 1, sd[f"{p}1.weight"], sd[f"{p}1.bias"], 1e-5))
    return f.permute(0, 2, 3, 1).reshape(B, T, H // 4, W // 4, -1)


def vmg_forward(sd: SD, cfg: VMGConfig, x: Tensor, mutate: bool = True, call_index: int = 1) -> Tensor:
    """VMG.forward (models/vmg.py:585-637): x (B,T,3,H,W) in [0,1] -> (B,T,3,4H,4W)."""
    B, T, C3, H, W = x.shape
    assert H >= 64 and W >= 64, "The height and width must larger than 64."
    up = F.interpolate(x.reshape(B * T, C3, H, W), scale_factor=4, mode="bilinear", align_corners=False)
    Hp, Wp = int(np.ceil(H / cfg.scale)) * cfg.scale, int(np.ceil(W / cfg.scale)) * cfg.scale
    xp = F.pad(x, (0, Wp - W, 0, Hp - H, 0, 0), mode="replicate")
    ff, fb = compute_flows(sd, cfg, xp)
    feat = F.leaky_relu(conv_nhwc(xp.reshape(B * T, C3, Hp, Wp).permute(0, 2, 3, 1), sd["input_proj.proj.0.weight"],
                                  sd["input_proj.proj.0.bias"], 1), 0.01)
    feat = feat.reshape(B, T, Hp, Wp, -1)

    ne, nd = cfg.num_enc_layers, cfg.num_dec_layers

    def enc(i, z):
        return mlp_encoder(sd, f"encoder_layers.{i}.", z, ff[i], fb[i], cfg, cfg.depths[i], cfg.chunk_h[i], cfg.chunk_w[i],
                           cfg.temporal_type[i], i, cfg.traj_res_n[i], cfg.num_heads[i], cfg.window_sizes[i], mutate, call_index)

    def dec(i, z):
        s = ne - 2 - i  # flow / chunk index of decoder stage i (models/vmg.py:337-355: index -i-2)
        return mlp_encoder(sd, f"decoder_layers.{i}.", z, ff[s], fb[s], cfg, cfg.depths[ne + i], cfg.chunk_h[-i - 2],
                           cfg.chunk_w[-i - 2], cfg.temporal_type[-i - 2], len(cfg.temporal_type) - i - 2,
                           cfg.traj_res_n[ne + i], cfg.num_heads[ne + i], cfg.window_sizes[ne + i], mutate, call_index)

    if cfg.num_layers > 3:  # forward_features_multi_stages, if_concat False (models/vmg.py:466-567)
        x1 = enc(0, feat)
        x1_3 = mdsc_skip(sd, "sc_64_16.", x1) if cfg.mdsc else 0
        x2 = enc(1, updown(sd, "downsample.0.", x1, "down"))
        x2_4 = mdsc_skip(sd, "sc_32_8.", x2) if cfg.mdsc else 0
        x3 = enc(2, updown(sd, "downsample.1.", x2, "down"))
        x4 = enc(3, updown(sd, "downsample.2.", x3 + x1_3, "down"))
        x5 = dec(0, updown(sd, "upsample.0.", x4 + x2_4, "up"))
        x6 = dec(1, updown(sd, "upsample.1.", x5 + x3, "up"))
        x7 = dec(2, updown(sd, "upsample.2.", x6 + x2, "up"))
        y = x7 + x1
    else:  # forward_features_few_stages (models/vmg.py:569-582)
        x1 = enc(0, feat)
        x2 = enc(1, updown(sd, "downsample.0.", x1, "down"))
        x3 = dec(0, updown(sd, "upsample.0.", x2, "up"))
        y = x3 + x1
    if cfg.if_local_fuse:
        y = feat + conv_nhwc(y.reshape(B * T, Hp, Wp, -1), sd["local_cnn.weight"], sd["local_cnn.bias"], 1).reshape(feat.shape)
    y = y[:, :, :H, :W].reshape(B * T, H, W, -1)
    out = sr_head(sd, y).permute(0, 3, 1, 2) + up
    return out.reshape(B, T, C3, 4 * H, 4 * W)


# --------------------------------------------------------------------------------------------------
# loss + metric (utils/loss.py:22-79, utils/metrics.py:11-26) -- used by the train-step parity tests
# --------------------------------------------------------------------------------------------------
def charbonnier_edge_loss(x: Tensor, y: Tensor, eps: float = 1e-12, aux_ratio: float = 0.005, aux: bool = True) -> Tensor:
    """CharbonnierLoss(+EdgeLoss) on (B,T,3,H,W).  utils/loss.py:32-42, 45-79."""
    loss = torch.mean(torch.sqrt((x - y) ** 2 + eps))
    if not aux:
        return loss
    k1 = torch.tensor([[.05, .25, .4, .25, .05]])
    kern = (k1.t() @ k1)[None].repeat(3, 1, 1, 1).to(x)

    def gauss(img):
        return F.conv2d(F.pad(img, (2, 2, 2, 2), mode="replicate"), kern, groups=3)

    def lap(img):
        f = gauss(img)
        z = torch.zeros_like(f)
        z[:, :, ::2, ::2] = f[:, :, ::2, ::2] * 4
        return img - gauss(z)

    T = x.shape[1]
    e = sum(torch.mean(torch.sqrt((lap(x[:, i]) - lap(y[:, i])) ** 2 + eps)) for i in range(T)) / T
    return loss + aux_ratio * e


def psnr_uint8(a: np.ndarray, b: np.ndarray) -> float:
    """PSNR of two images in [0,255] (utils/metrics.py:11-26)."""
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    if mse == 0:
        return float("inf")
    return 20 * math.log10(255.0 / math.sqrt(mse))
