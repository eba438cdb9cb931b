"""VMG network with the reference's nn.Module surface (models/vmg.py:176-637) over the HIP kernels.

Drop-in contract: same constructor keywords, same attribute names read by the reference's
Trainer/Tester (``spynet``, ``mlp_wd_param``, ``num_out_frames``), same state-dict keys / shapes / layouts (conv
OIHW, Linear (out,in), buffers gamma_h/gamma_w/decay_v/relative_position_index/spynet.mean/std), same call
``module(x, flow_pretrained=None, config_amp=None)`` with x (B,T,3,H,W) -> (B,T,3,4H,4W).

MI355X-first differences from the reference's Python (none change results):
  * features live channels-last (B*T, H, W, C) end to end: no NCHW<->NHWC round trips around every conv;
  * every convolution / Linear is the MFMA implicit-GEMM kernel with bias, activation, residual, scale and
    PixelShuffle fused in its epilogue; channel concats are virtual (several source pointers);
  * nn.Conv2d / nn.Linear / nn.LayerNorm objects below are PARAMETER HOLDERS (they give identical state-dict
    keys and initialisation); their own forward() is never used on the hot path.

There is no CPU path: calling the module on CPU tensors raises (vmg_amd.hip.HipError).
"""
from __future__ import annotations

from fractions import Fraction
from typing import List, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as FH
from . import hip
from .hip import ACT_GELU, ACT_LRELU, ACT_RELU, HipError


# ---------------------------------------------------------------------------------------------------------
# small helpers
# ---------------------------------------------------------------------------------------------------------
def conv(mod: nn.Conv2d, srcs: Sequence[torch.Tensor], N: int, H: int, W: int, **kw) -> torch.Tensor:
    return FH.conv2d(srcs, mod.weight, mod.bias, N, H, W, ks=mod.kernel_size[0], **kw)


def lin(mod: nn.Linear, x: torch.Tensor, **kw) -> torch.Tensor:
    return FH.linear(x, mod.weight, mod.bias, **kw)


def lnorm(mod: nn.LayerNorm, x: torch.Tensor) -> torch.Tensor:
    return FH.layer_norm(x, mod.weight, mod.bias, mod.eps)


def flow_warp_nhwc(x: torch.Tensor, flow: torch.Tensor, mode: str = "bilinear", padding: str = "zeros") -> torch.Tensor:
    """x (n,h,w,c) sampled at pixel grid + flow (n,h,w,2), align_corners=True (models/trajectory.py:71-116)."""
    n, h, w, c = x.shape
    if (h, w) != tuple(flow.shape[1:3]):
        raise ValueError(f"The spatial sizes of input ({(h, w)}) and flow ({tuple(flow.shape[1:3])}) are not the same.")
    return FH.grid_sample_flow(x, flow, mode, padding)


# ---------------------------------------------------------------------------------------------------------
# SPyNet (models/vmg.py:18-173) on the HIP kernels: 7x7 convolutions = the implicit-GEMM kernel (KS = 7) with fused ReLU,
# pyramid = vmg_avgpool2_nhwc, flow up-sampling = vmg_upsample2x_ac, warps = vmg_warp_bilinear.  Channels-last throughout;
# the flow itself (up-sampling, warp coordinates, the residual accumulation over levels) stays fp32 in every compute dtype.
# ---------------------------------------------------------------------------------------------------------
class _ConvModule(nn.Module):  # mmcv ConvModule naming: `.conv` holds the parameters (state-dict keys ...conv.weight / ...conv.bias)
    def __init__(self, cin, cout, act):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 7, 1, 3)
        self.act = act


class SPyNetBasicModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.basic_module = nn.Sequential(_ConvModule(8, 32, True), _ConvModule(32, 64, True), _ConvModule(64, 32, True),
                                          _ConvModule(32, 16, True), _ConvModule(16, 2, False))

    def forward(self, srcs: Sequence[torch.Tensor], inner_dtype=None) -> torch.Tensor:
        """srcs: channels-last tensors whose channels concatenate to the 8 input channels [ref, warped, flow] -> (n,h,w,2).
        inner_dtype: the three middle convolutions (32 -> 64 -> 32 -> 16, 94 % of the module's FLOPs) compute in this dtype while the first and
        the last one -- the two that touch the flow -- keep the dtype of `srcs` (SPyNet.edge_fp32)."""
        n, h, w = srcs[0].shape[:3]
        y = list(srcs)
        last = len(self.basic_module) - 1
        for i, m in enumerate(self.basic_module):
            if inner_dtype is not None and i == 1:
                y = [y[0].to(inner_dtype)]
            if inner_dtype is not None and i == last:
                y = [y[0].to(srcs[0].dtype)]
            y = [FH.conv2d(y, m.conv.weight, m.conv.bias, n, h, w, ks=7, act=ACT_RELU if m.act else hip.ACT_NONE, fuse_src_act=i > 0)]
        return y[0]


class SPyNet(nn.Module):
    def __init__(self, pretrained=None):
        super().__init__()
        self.basic_module = nn.ModuleList([SPyNetBasicModule() for _ in range(6)])
        if isinstance(pretrained, str):
            import os
            if os.path.isfile(pretrained):
                sd = torch.load(pretrained, map_location="cpu")
                self.load_state_dict(sd.get("state_dict", sd), strict=True)
            else:
                import warnings
                warnings.warn(f"SPyNet weights '{pretrained}' are not a local file (no network here): keeping random init")
        elif pretrained is not None:
            raise TypeError(f"[pretrained] should be str or None, but got {type(pretrained)}.")
        self.register_buffer("mean", torch.Tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer("std", torch.Tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))
        # bf16 runs: everything that touches the 2-channel flow -- the warp of the support frame, the 8-channel operand [ref, warped, flow], the
        # first convolution (8 -> 32: its data gradient IS the flow gradient of the level below) and the last one (16 -> 2: its output is the
        # flow residual, its output gradient the flow gradient) -- stays fp32; only the three middle convolutions run in bf16.
        self.edge_fp32 = False

    def compute_flow(self, ref: torch.Tensor, supp: torch.Tensor) -> torch.Tensor:
        """ref, supp: channels-last (n,h,w,8) in the compute dtype, normalised RGB in channels 0..2 and zeros behind (the warp
        kernel moves 16-byte channel vectors), h and w multiples of 32 -> flow (n,h,w,2) fp32 (models/vmg.py:39-85)."""
        from . import kernels as K
        n, h, w, _ = ref.shape
        inner = None
        if self.edge_fp32 and ref.dtype != torch.float32:
            inner, ref, supp = ref.dtype, ref.float(), supp.float()
        dt = ref.dtype
        refs, supps = [ref], [supp]
        with torch.no_grad():  # the pyramid of the input frames needs no gradient
            for _ in range(5):
                refs.append(K.avgpool2(refs[-1]))
                supps.append(K.avgpool2(supps[-1]))
        refs, supps = refs[::-1], supps[::-1]
        flow = torch.zeros(n, h // 32, w // 32, 2, dtype=torch.float32, device=ref.device)
        for level in range(6):
            up = flow if level == 0 else FH.upsample2x_flow(flow, 2.0)
            warped = FH.grid_sample_flow(supps[level], up, "bilinear", "border")
            # one 8-channel operand [ref, warped, flow] (a 16-byte vector per pixel) instead of a virtual concat of 3 + 3 + 2 channels:
            # the convolution and its weight gradient then move whole vectors
            x8 = FH.spy_operand(refs[level], warped, up)
            res = self.basic_module[level]([x8], inner)
            flow = FH.spy_flow_add(up, res)
        return flow

    def forward(self, ref, supp, compute_dtype=torch.float32):
        """ref, supp (n,3,h,w) in [0,1] -> flow (n,2,h,w) from supp to ref, fp32 (models/vmg.py:87-123)."""
        hip.require_cuda(ref, supp)
        h, w = ref.shape[2:4]
        w_up = w if (w % 32) == 0 else 32 * (w // 32 + 1)
        h_up = h if (h % 32) == 0 else 32 * (h // 32 + 1)
        resized = (h_up, w_up) != (h, w)

        def prep(img):
            from . import kernels as K
            if not resized and not img.requires_grad and compute_dtype in (torch.float32, torch.bfloat16):
                return K.spy_prep(img.float().contiguous(), self.mean.float(), self.std.float(), compute_dtype)  # normalise + channels-last + 3 -> 8 channels + cast: one kernel
            img = (img.float() - self.mean) / self.std
            if resized:  # (sizes that are no multiple of 32: a plain bilinear resize, models/vmg.py:104-113)
                img = F.interpolate(img, size=(h_up, w_up), mode="bilinear", align_corners=False)
            return F.pad(img.permute(0, 2, 3, 1), (0, 5)).to(compute_dtype).contiguous()  # channels-last, 3 -> 8 channels

        flow = self.compute_flow(prep(ref), prep(supp)).permute(0, 3, 1, 2)
        if resized:
            flow = F.interpolate(flow, size=(h, w), mode="bilinear", align_corners=False)
            flow = torch.stack((flow[:, 0] * (float(w) / float(w_up)), flow[:, 1] * (float(h) / float(h_up))), 1)
        return flow


# ---------------------------------------------------------------------------------------------------------
# TAB block (models/function.py)
# ---------------------------------------------------------------------------------------------------------
class _CALayer(nn.Module):
    def __init__(self, channel, reduction):
        super().__init__()
        self.conv_du = nn.Sequential(nn.Conv2d(channel, channel // reduction, 1), nn.ReLU(), nn.Conv2d(channel // reduction, channel, 1),
                                     nn.Sigmoid())


class RCAB(nn.Module):
    """conv3x3+ReLU, conv3x3, channel attention, + input (models/function.py:561-583, 542-558)."""

    def __init__(self, n_feat, reduction=8):
        super().__init__()
        self.body = nn.Sequential(nn.Conv2d(n_feat, n_feat, 3, 1, 1), nn.ReLU(), nn.Conv2d(n_feat, n_feat, 3, 1, 1),
                                  _CALayer(n_feat, reduction))

    def forward(self, x, out_scale: float = 1.0, x_res=None):
        """x_res: a second autograd handle of x for the residual add (functional.layer_norm_fan: the two gradients w.r.t. x are then summed
        by the LayerNorm backward kernel instead of by an add pass of their own)."""
        B, T, H, W, C = x.shape
        N = B * T
        r = conv(self.body[0], [x], N, H, W, act=ACT_RELU)
        r = conv(self.body[2], [r], N, H, W, fuse_src_act=True)  # (the ReLU's derivative rides on this conv's data-gradient launch)
        du = self.body[3].conv_du
        return FH.channel_attention_residual(r, (x if x_res is None else x_res).reshape(N, H, W, C), du[0].weight, du[0].bias, du[2].weight, du[2].bias,
                                             out_scale).reshape(B, T, H, W, C)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward(self, x):
        return lin(self.fc2, lin(self.fc1, x, act=ACT_GELU), fuse_src_act=True)


class Mlp_cnn(nn.Module):
    """conv3x3(C->rC, groups)+GELU then Linear(rC->C) (models/function.py:50-79)."""

    def __init__(self, in_features, exp_r=4, n_groups=1):
        super().__init__()
        self.hidden_features = int(in_features * exp_r)
        self.n_groups = n_groups
        self.fc1 = nn.Conv2d(in_features, self.hidden_features, 3, 1, 1, groups=n_groups)
        self.fc2 = nn.Linear(self.hidden_features, in_features)

    def forward(self, x, res=None):
        B, T, H, W, C = x.shape
        N = B * T
        if self.n_groups == 1:
            h = conv(self.fc1, [x], N, H, W, act=ACT_GELU)
        else:
            h = FH.grouped_conv2d(x, self.fc1.weight, self.fc1.bias, self.n_groups, N, H, W, ks=3, act=ACT_GELU)
        return lin(self.fc2, h, res=res, fuse_src_act=True).reshape(B, T, H, W, C)


class Enhanced_MorphFCs_decay(nn.Module):
    """Token mixer: H-, W- and channel-branch, softmax re-weighting, projection, tanh gate
    (models/function.py:596-805).  STATEFUL like the reference: mlp_h/mlp_w weights are multiplied by
    gamma_h/gamma_w in place on every forward call (SURVEY trap T1)."""

    def __init__(self, dim, chunk_h=8, chunk_w=8, qkv_bias=False, channel_mixer="rcab"):
        super().__init__()
        self.chunk_h, self.chunk_w = chunk_h, chunk_w
        self.Ch = int(np.ceil(dim / chunk_h)) * chunk_h
        self.Cw = int(np.ceil(dim / chunk_w)) * chunk_w
        self.mlp_h = nn.Sequential(nn.Linear(self.Ch, self.Ch, bias=qkv_bias), nn.ReLU())
        self.mlp_w = nn.Sequential(nn.Linear(self.Cw, self.Cw, bias=qkv_bias), nn.ReLU())
        if channel_mixer == "rcab":
            self.mlp_c = RCAB(n_feat=dim)
        elif channel_mixer == "vanilla":  # VMG.__init__'s default, what VMG-REDS.yml (no such key) gets: Linear + ReLU (function.py:640-644)
            self.mlp_c = nn.Sequential(nn.Linear(dim, dim, bias=qkv_bias), nn.ReLU())
        else:
            raise NotImplementedError(f"channel_mixer {channel_mixer!r}: 'rcab' or 'vanilla'")
        self.reweight = Mlp(dim, dim // 4, dim * 3)
        self.proj = nn.Linear(dim, dim)
        self.register_buffer("gamma_h", decay_gamma(chunk_h, self.Ch))
        self.register_buffer("gamma_w", decay_gamma(chunk_w, self.Cw))

    N_HANDLES = 5  # consumers of the mixer's input: H branch, W branch, channel branch (RCAB: conv + residual), tanh gate

    def forward(self, x, residual=None):
        """x: the normalised features (B,T,H,W,C), or a list of N_HANDLES autograd handles of them (functional.layer_norm_fan).
        residual = (res, drop_prob, training, scale): return res + DropPath(mixer(x)) * scale instead of mixer(x) -- the TAB residual taken
        inside the gate's pass (functional.gate_residual)."""
        xs = list(x) if isinstance(x, (list, tuple)) else [x] * self.N_HANDLES
        x = xs[0]
        B, T, H, W, C = x.shape
        if getattr(self, "_t1_done", False):  # (VMG.forward has applied this call's decay for all mixers at once, FH.decay_weights_and_repack)
            self._t1_done = False
        else:
            with torch.no_grad():  # T1: persistent, outside autograd, before the GEMM (function.py:766-768, 779-781)
                self.mlp_h[0].weight.mul_(self.gamma_h)
                self.mlp_w[0].weight.mul_(self.gamma_w)
        # token reshuffle + Linear + ReLU + 1/Ch + inverse reshuffle: one kernel per branch where it is instantiated
        h = FH.morph_linear(xs[0], self.mlp_h[0].weight, self.mlp_h[0].bias, "h", self.chunk_h, self.Ch)
        w = FH.morph_linear(xs[1], self.mlp_w[0].weight, self.mlp_w[0].bias, "w", self.chunk_w, self.Cw)
        if isinstance(self.mlp_c, RCAB):
            c = self.mlp_c(xs[2], out_scale=1.0 / C, x_res=xs[3])
        else:
            c = lin(self.mlp_c[0], xs[2], act=ACT_RELU, alpha=1.0 / C)
        rw = self.reweight
        y = FH.reweight_mix(h, w, c, rw.fc1.weight, rw.fc1.bias, rw.fc2.weight, rw.fc2.bias)
        y = lin(self.proj, y)
        if residual is not None:
            return FH.gate_residual(xs[4], y, *residual)
        return FH.tanh_gate(xs[4], y)


def decay_gamma(chunk: int, ch_total: int) -> torch.Tensor:
    """(Ch,Ch) retention matrix of form_decay (models/function.py:651-652, 684-732) in closed form:
    Gamma[r, c] = mean_i d_i^(|r//S - c//S| + 1), d_i = exp(log(1 - 2^-(5 + chunk-1-i))), S = Ch / chunk,
    powers by repeated multiplication (the reference's cumulative product)."""
    S = ch_total // chunk
    d = torch.log(1 - 2 ** (-5 - torch.arange(chunk - 1, -1, -1, dtype=torch.float))).exp()
    powers = [d.clone()]
    for _ in range(1, chunk):
        powers.append(powers[-1] * d)
    powers = torch.stack(powers, 1)
    idx = torch.arange(chunk)
    blk = powers[:, (idx[:, None] - idx[None, :]).abs()]
    full = blk[:, :, None, :, None].expand(chunk, chunk, S, chunk, S).reshape(chunk, chunk * S, chunk * S)
    return torch.mean(full, 0)


class TAB(nn.Module):
    """x += DP(spatial(LN2(x))) * s; x += DP(ffn(LN3(x))) * s (models/function.py:1139-1217)."""

    def __init__(self, embed_dim, chunk_h, chunk_w, mlp_ratio, n_groups, qkv_bias, drop_path, ffn, mixer_scaling, channel_mixer):
        super().__init__()
        self.spatial_scale = mixer_scaling
        self.norm2 = nn.LayerNorm(embed_dim)
        self.spatial_mixing = Enhanced_MorphFCs_decay(embed_dim, chunk_h, chunk_w, qkv_bias, channel_mixer)
        self.norm3 = nn.LayerNorm(embed_dim)
        if ffn == "vanilla":
            self.channel_mixing = Mlp(embed_dim, int(embed_dim * mlp_ratio))
        elif ffn == "ffn_cnn":
            self.channel_mixing = Mlp_cnn(embed_dim, exp_r=mlp_ratio, n_groups=n_groups)
        else:
            raise NotImplementedError(f"ffn_type {ffn!r} is not used by any shipped config")
        self.drop_prob = float(drop_path)

    def forward(self, x):
        s = self.spatial_scale
        dp = self.drop_prob > 0.0 and self.training
        # x feeds the norm AND the residual: layer_norm_skip hands x back so that both gradients meet in the LayerNorm backward kernel
        # ... and the mixer reads the normalised tensor five times: layer_norm_fan hands out one autograd handle per consumer, so that ALL their
        # gradients (and the skip gradient) are summed inside the one LayerNorm backward kernel
        if torch.is_grad_enabled() and x.requires_grad:
            n2, xs = FH.layer_norm_fan(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, Enhanced_MorphFCs_decay.N_HANDLES)
        else:
            n2, xs = FH.layer_norm_skip(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        x = self.spatial_mixing(n2, residual=(xs, self.drop_prob, self.training, s))  # gate, DropPath mask, scale and residual add: one pass
        n3, xs = FH.layer_norm_skip(x, self.norm3.weight, self.norm3.bias, self.norm3.eps)
        if isinstance(self.channel_mixing, Mlp_cnn) and not dp and s == 1.0:
            return self.channel_mixing(n3, res=xs)  # residual fused in the Linear epilogue
        return FH.residual_drop_path(xs, self.channel_mixing(n3), self.drop_prob, self.training, s)


# ---------------------------------------------------------------------------------------------------------
# trajectory attention + recurrent residual chains (models/trajectory.py)
# ---------------------------------------------------------------------------------------------------------
class ResidualBlockNoBN0(nn.Module):
    def __init__(self, mid_channels, res_scale):
        super().__init__()
        self.res_scale = res_scale
        self.conv1 = nn.Conv2d(mid_channels, mid_channels, 3, 1, 1)
        self.conv2 = nn.Conv2d(mid_channels, mid_channels, 3, 1, 1)


class ResidualBlocksWithInputConv(nn.Module):
    """conv3x3(2C->C)+LeakyReLU(0.1), then num_blocks x [conv+ReLU, conv, x + r*out] (trajectory.py:16-52,165-221)."""

    def __init__(self, in_channels, out_channels, num_blocks, r_scaling):
        super().__init__()
        self.main = nn.Sequential(nn.Conv2d(in_channels, out_channels, 3, 1, 1), nn.LeakyReLU(0.1),
                                  nn.Sequential(*[ResidualBlockNoBN0(out_channels, r_scaling) for _ in range(num_blocks)]))
        self.recompute = False  # VMG(recompute_chains=True): drop the 2 * num_blocks intermediates, rebuild them in the backward
        self.fp8 = False        # VMG(fp8_chains=True): the block convolutions of the forward on the fp8 kernel

    def forward(self, srcs: Sequence[torch.Tensor]):
        blocks = list(self.main[2])
        r = blocks[0].res_scale if blocks else 1.0
        return FH.residual_chain([t.contiguous() for t in srcs], self.main[0], blocks, r, recompute=self.recompute and torch.is_grad_enabled(), fp8=self.fp8)


class LTAM_multi_head(nn.Module):
    """Window trajectory attention (trajectory.py:493-547, 672-795), mode 'wins'."""

    def __init__(self, embed_dim, head, if_scale, twins):
        super().__init__()
        self.head = head
        self.scale = (embed_dim // head) ** -0.5 if if_scale else 1.0
        self.proj = nn.Linear(embed_dim, embed_dim)
        self.win_h, self.win_w = twins
        self.relative_pos_encoding = nn.Parameter(torch.zeros(head, self.win_h * self.win_w, self.win_h * self.win_w))
        nn.init.trunc_normal_(self.relative_pos_encoding, std=.02)
        self.register_buffer("decay_v", 1 - 2 ** (-5 - torch.arange(head - 1, -1, -1, dtype=torch.float32)))

    def forward(self, q, keys, anchor, vals, loc):
        """q, anchor (n,h,w,c); keys, vals: lists of t tensors (n,h,w,c); loc (n,2t,h,w) -> (n,h,w,c)."""
        att = FH.ltam_attention(q, keys, vals, loc, self.relative_pos_encoding, self.decay_v, self.head, self.win_h, self.win_w,
                                self.scale)
        return lin(self.proj, att, res=anchor)


class Trajectory_multi_head(nn.Module):
    """Bidirectional recurrence (trajectory.py:226-490): warp state by flow, attend to key-frame memory at tracked
    locations, then the residual conv chain; finally fuse [backward, input, forward] with a 1x1 conv."""

    def __init__(self, embed_dim, num_blocks, frame_stride, head, head_scale, r_scaling, twins):
        super().__init__()
        self.embed_dims = embed_dim
        self.keyframe_stride = frame_stride
        self.LTAM = LTAM_multi_head(embed_dim, head, head_scale, twins)
        self.resblocks = ResidualBlocksWithInputConv(2 * embed_dim, embed_dim, num_blocks, r_scaling)
        self.fusion = nn.Conv2d(3 * embed_dim, embed_dim, 1, 1, 0)

    def forward(self, x, flows_forward, flows_backward):
        """The two direction sweeps of the reference (trajectory.py:323-392 backward, :407-477 forward) share LTAM and
        resblocks and do not depend on each other, and at sweep step j both hold the same number of key frames
        (j % stride == 0 marks a key frame in either direction).  They therefore run in LOCKSTEP as one batch of 2n
        frames -- rows [0, n) are the backward sweep at frame t-1-j, rows [n, 2n) the forward sweep at frame j -- which
        doubles the pixels per launch of every kernel of the recurrence (M = 2*n*h*w) and halves the launch count."""
        n, t, h, w, c = x.shape
        fb = flows_backward.permute(1, 0, 3, 4, 2).float()
        ff = flows_forward.permute(1, 0, 3, 4, 2).float()
        s = self.keyframe_stride
        x_steps, x = FH.fan_out(x, 2)  # (the features feed the recurrence AND the fusion conv below)
        curs = FH.pair_frame_steps(x_steps)  # t tensors (2n, h, w, c): step j -> [frame t-1-j | frame j], one gather from the batch-major features
        # step j >= 1 warps by flows_backward[:, t-1-j] (backward sweep) and flows_forward[:, j-1] (forward sweep)
        flpair = torch.cat([fb.flip(0), ff], 1).contiguous() if t > 1 else None  # (t-1, 2n, h, w, 2); row j-1 serves step j
        ident = FH.identity_grid(2 * n, h, w, x.device)
        loc = ident
        feat = None
        k_in: List[torch.Tensor] = []
        k_state: List[torch.Tensor] = []
        feats = []
        # (unbind instead of indexing: the backward of fl[j] is a zero-filled tensor of the WHOLE stack per frame, summed pairwise by autograd; the
        # backward of unbind is one stack.  The features' step tensors come from / go to one kernel each way: pair_frame_steps, unpair_steps.)
        fls = flpair.unbind(0) if flpair is not None else ()
        for j in range(t):
            key, last = j % s == 0, j == t - 1
            # a step's input frame feeds the attention (as its query), the residual chain and -- on a key frame -- the attention memory; the hidden state feeds
            # the output, the next step's warp and the memory: one handle per consumer (FH.fan_out), so that the gradient is ONE sum, not autograd's pairwise adds
            hd = list(FH.fan_out(curs[j], (1 if j > 0 else 0) + 1 + (1 if key else 0)))
            cur_q = hd.pop() if j > 0 else None
            cur_c = hd.pop()
            cur_k = hd.pop() if key else None
            if j == 0:
                feat = torch.zeros_like(cur_c)
            else:
                fl = fls[j - 1]
                feat = flow_warp_nhwc(feat_next, fl, "bilinear", "border")
                loc = FH.warp_locations(loc, fl)
                feat = self.LTAM(cur_q, k_in, feat, k_state, loc)
                if key:
                    loc = torch.cat([loc, ident], 1)
            feat = self.resblocks([cur_c, feat])
            hd = list(FH.fan_out(feat, 1 + (0 if last else 1) + (1 if key else 0)))
            f_out = hd.pop()
            feat_next = None if last else hd.pop()
            if key:  # (grad_bank: the later frames' attention backward calls sum their gradients w.r.t. this key-frame in one buffer)
                k_state.append(FH.grad_bank(hd.pop()))
                k_in.append(FH.grad_bank(cur_k))
            feats.append(f_out)
        # step j: (backward sweep at frame t-1-j, forward sweep at frame j) -> both sweeps (n, t, h, w, c) in frame order.  The fusion is a 1x1 conv:
        # its pixels may come in any order -- batch-major like x, so neither x nor the result is transposed
        back, fwd = FH.unpair_steps(feats, n)
        out = conv(self.fusion, [back, x, fwd], n * t, h, w, act=ACT_LRELU, slope=0.1)
        return out.reshape(n, t, h, w, c)


# ---------------------------------------------------------------------------------------------------------
# 3-D shifted-window attention (models/swin_3d.py), only with temporal_empty=False
# ---------------------------------------------------------------------------------------------------------
def _get_window_size(x_size, window_size, shift_size):
    ws, ss = list(window_size), list(shift_size)
    for i in range(3):
        if x_size[i] <= window_size[i]:
            ws[i] = x_size[i]
            ss[i] = 0
    return tuple(ws), tuple(ss)


def _relative_position_index(ws) -> torch.Tensor:
    coords = torch.stack(torch.meshgrid(torch.arange(ws[0]), torch.arange(ws[1]), torch.arange(ws[2]), indexing="ij"))
    cf = coords.flatten(1)
    rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws[0] - 1
    rel[:, :, 1] += ws[1] - 1
    rel[:, :, 2] += ws[2] - 1
    rel[:, :, 0] *= (2 * ws[1] - 1) * (2 * ws[2] - 1)
    rel[:, :, 1] *= 2 * ws[2] - 1
    return rel.sum(-1)


class rWindowAttention(nn.Module):
    """Each time slice's queries attend to the tokens of the other slices of the window (swin_3d.py:120-252)."""

    def __init__(self, dim, window_size, num_heads, qkv_bias):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, tuple(window_size), num_heads
        n_rel = (2 * window_size[0] - 1) * (2 * window_size[1] - 1) * (2 * window_size[2] - 1)
        self.relative_position_bias_table = nn.Parameter(torch.zeros(n_rel, num_heads))
        self.register_buffer("relative_position_index", _relative_position_index(window_size))
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, 2 * dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def forward(self, y, shift):
        """y (B,D,H,W,C): the normalised, UN-partitioned feature map; shift: the block's (sd, sh, sw) (zeros when not shifted)."""
        q = lin(self.q, y)
        kv = lin(self.kv, y)
        o = FH.win3d_attention(q, kv, self.q.bias, self.kv.bias, self.relative_position_bias_table, self.num_heads, self.window_size[0], shift)
        return o


class EncoderBlockOnOnetoken(nn.Module):
    def __init__(self, dim, num_heads, window_size, shift_size, mlp_ratio, qkv_bias):
        super().__init__()
        self.window_size, self.shift_size, self.num_heads = tuple(window_size), tuple(shift_size), num_heads
        self.norm1 = nn.LayerNorm(dim)
        self.attn = rWindowAttention(dim, window_size, num_heads, qkv_bias)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        B, D, H, W, C = x.shape
        ws, ss = _get_window_size((D, H, W), self.window_size, self.shift_size)
        if tuple(ws) != (self.window_size[0], 8, 8):
            raise HipError(f"3-D window attention: the feature map {(D, H, W)} must be at least one (wt, 8, 8) window ({self.window_size})")
        y = lnorm(self.norm1, x)
        # window partition, padding, roll and mask live in the attention kernel's addressing; proj adds the residual in its epilogue
        x = lin(self.attn.proj, self.attn(y, ss), res=x)
        z = lnorm(self.norm2, x)
        return lin(self.mlp.fc2, lin(self.mlp.fc1, z, act=ACT_GELU), res=x)


class DecoderLayer(nn.Module):
    """swin_3d.DecoderLayer (swin_3d.py:1108-1202): depth blocks, odd ones shifted by window/2."""

    def __init__(self, dim, depth, num_heads, window_size, mlp_ratio, qkv_bias):
        super().__init__()
        self.window_size = list(window_size)
        self.shift_size = [i // 2 for i in window_size]
        self.blocks = nn.ModuleList([EncoderBlockOnOnetoken(dim, num_heads, window_size, [0, 0, 0] if i % 2 == 0 else self.shift_size,
                                                            mlp_ratio, qkv_bias) for i in range(depth)])

    def forward(self, x):
        B, D, H, W, C = x.shape
        wt = self.window_size[0]
        seq_back = None
        if D % wt != 0:  # rearrange_shape: repeat the last full-window frames in reverse (swin_3d.py:1141-1157)
            delta_t = int(np.ceil(D / wt)) * wt - D
            delta = list(range(-1, -(delta_t + 1), -1))
            start = list(range(0, (D // wt) * wt))
            end = list(range((D // wt) * wt, D))
            new_seq = start + delta + end
            seq_back = start + list(range(-1, -(len(end) + 1), -1))[::-1]
            rep = [start[i] for i in delta]
            x = torch.cat([x, x[:, rep]], 1)[:, new_seq].contiguous()
            D = x.shape[1]
        for blk in self.blocks:
            x = blk(x)
        if seq_back is not None:
            x = x[:, seq_back].contiguous()
        return x


# ---------------------------------------------------------------------------------------------------------
# stage container, sampling, whole network
# ---------------------------------------------------------------------------------------------------------
class Mlp_encoder(nn.Module):
    """TAB stack + local conv residual + temporal module (models/function.py:1267-1543); channels-last in/out."""

    def __init__(self, embed_dim, depth, segm, chunk_dim_h, chunk_dim_w, mlp_ratio, n_groups, qkv_bias, drop_path, window_size,
                 n_nonkeyframes, aligned, empty_aligned, traj_r_n, traj_heads, if_smooth, region_range, ffn_type, r_scaling, twins,
                 traj_scale, m_scaling, if_local_fuse, channel_mixer):
        super().__init__()
        self.aligned, self.empty = aligned, empty_aligned
        self.if_smooth, self.region_range = if_smooth, region_range
        self.local_fuse = if_local_fuse
        if if_local_fuse:
            self.local_cnn = nn.Conv2d(embed_dim, embed_dim, 3, 1, 1)
        self.mlp_blocks = nn.ModuleList([TAB(embed_dim, chunk_dim_h, chunk_dim_w, mlp_ratio, n_groups, qkv_bias,
                                             drop_path[i] if isinstance(drop_path, list) else drop_path, ffn_type, m_scaling,
                                             channel_mixer) for i in range(depth)])
        if aligned:
            raise NotImplementedError("temporal_type=True (DCN alignment) is not configured by any shipped config")
        if aligned is None:
            self.traj_mixing = nn.Identity() if empty_aligned else DecoderLayer(embed_dim, 2, segm, window_size, mlp_ratio, qkv_bias)
        else:
            self.traj_mixing = Trajectory_multi_head(embed_dim, traj_r_n, n_nonkeyframes, traj_heads, traj_scale, r_scaling, twins)

    @staticmethod
    def flow_smoothing(flow, r):
        """reflect-pad to a multiple of r, r x r mean, nearest x r, crop (function.py:1466-1478)."""
        H, W = flow.shape[-2:]
        hf, wf = int(np.ceil(H / r)) * r, int(np.ceil(W / r)) * r
        if flow.is_cuda and hf - H < H and wf - W < W:
            return FH.flow_smooth(flow, r)  # one kernel each way (csrc/warp.hip)
        B, T, C = flow.shape[:3]
        f = flow.reshape(-1, C, H, W)
        f = F.pad(f, (0, wf - W, 0, hf - H), mode="reflect")
        f = F.adaptive_avg_pool2d(f, (hf // r, wf // r))
        n_, c_, hq, wq = f.shape
        f = f[:, :, :, None, :, None].expand(n_, c_, hq, r, wq, r).reshape(n_, c_, hq * r, wq * r)[..., :H, :W].contiguous()
        return f.view(B, T, C, H, W)

    def forward(self, x, flow_forward=None, flow_backward=None):
        B, T, H, W, C = x.shape
        shortcut = x
        if flow_forward is not None and self.if_smooth:
            flow_backward = self.flow_smoothing(flow_backward, self.region_range)
            flow_forward = self.flow_smoothing(flow_forward, self.region_range)
        for blk in self.mlp_blocks:
            x = blk(x)
        if self.local_fuse:
            x = conv(self.local_cnn, [x], B * T, H, W, res=shortcut.reshape(B * T, H, W, C)).reshape(B, T, H, W, C)
        if self.aligned is None:
            if not self.empty:
                x = self.traj_mixing(x)
        else:
            x = self.traj_mixing(x, flow_forward, flow_backward)
        return x


class UpdownkeepSampling(nn.Module):
    """space<->depth with channel order (neiw neih c), LayerNorm, Linear (models/layers.py:761-798)."""

    def __init__(self, dim_in, dim_out, mode):
        super().__init__()
        self.mode = mode
        cn = dim_in * 4 if mode == "down" else dim_in // 4
        self.norm = nn.LayerNorm(cn)
        self.linear = nn.Linear(cn, dim_out)

    def forward(self, x):
        # rearrangement + LayerNorm are one kernel (the rows are gathered from x); the Linear consumes the normalized rows
        return lin(self.linear, FH.space_depth_layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps, self.mode))


class InputProj(nn.Module):
    def __init__(self, in_channels, embed_dim):
        super().__init__()
        self.proj = nn.Sequential(nn.Conv2d(in_channels, embed_dim, 3, 1, 1), nn.LeakyReLU(inplace=True))


class VMG(nn.Module):
    def __init__(self, in_chans=3, embed_dim=[112, 224, 224, 448, 448, 224, 224, 112], depths=[8, 8, 8, 8, 8, 8, 8, 8],
                 num_heads=[2, 4, 8, 16, 16, 8, 4, 2], num_frames=7,
                 window_sizes=[(4, 4), (4, 4), (4, 4), (4, 4), (4, 4), (4, 4), (4, 4), (4, 4)], mdsc=True, if_concat=False,
                 mlp_ratio=4., n_groups=1, qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1,
                 norm_layer=nn.LayerNorm, patch_norm=True, back_RBs=0, spynet_pretrained=None, image_size=[64, 112], is_train=True,
                 if_print=False, ltam=True, traj_win=[16, None, None, None], traj_keyframes_n=[3, None, None, None],
                 traj_heads=[4, None, None, None], temporal_type=[False, None, None, None], temporal_empty=True,
                 traj_res_n=[1, 0, 0, 0, 0, 0, 1], deform_groups=[8, 16, 16, 32], max_residual_scale=[1, 2, 2, 4],
                 spatial_type=[False, False, False, False], flow_smooth=True, smooth_region_range=4, retention_decay=True,
                 non_linear=True, gating=True, symm=True, symm_act=nn.Tanh, relu_scale=True, relu_scale_norm=False,
                 ffn_type='vanilla', mixer_type=['mbconv', 'mbconv', 'mlps', 'mlps'], mixer_n=[2, 3, None, None], r_scaling=1.,
                 chunk_ratios=[1 / 4, 1 / 4, 3 / 16, 1 / 8], traj_mode='wins', twins=[2, 2], traj_scale=True, traj_refine=None,
                 m_scaling=1., if_local_fuse=False, channel_mixer='vanilla', compute_dtype=torch.float32, recompute_chains=False, fp8_chains=False):
        super().__init__()
        # --- options the hand-written path implements (every self-consistent shipped config)
        if not (ltam and retention_decay and non_linear and gating and symm and relu_scale) or relu_scale_norm or if_concat:
            raise NotImplementedError("VMG HIP path: ltam/retention_decay/non_linear/gating/symm/relu_scale must be true, "
                                      "relu_scale_norm/if_concat false (as in all shipped configs)")
        if symm_act not in ("tanh", nn.Tanh) or traj_mode != "wins" or any(m != "mlps" for m in mixer_type[:len(depths) // 2 + 1]):
            raise NotImplementedError("VMG HIP path: symm_act='tanh', traj_mode='wins', mixer_type='mlps' only")
        if back_RBs != 0 or traj_refine is not None:
            raise NotImplementedError("back_RBs > 0 / traj_refine are not used by any shipped config")
        self.num_layers = len(depths)
        self.num_enc_layers = self.num_layers // 2 + 1
        self.num_dec_layers = self.num_layers // 2
        self.scale = 2 ** (self.num_enc_layers - 1)
        dec_depths = depths[self.num_enc_layers:]
        self.embed_dim = embed_dim
        self.num_in_frames = num_frames
        self.num_out_frames = num_frames
        self.is_train = is_train
        self.if_print = if_print
        self.init_H, self.init_W = image_size
        self.compute_dtype = compute_dtype
        self.spynet_dtype = None  # None: SPyNet computes in compute_dtype; torch.float32: the flow network alone in fp32 (tools/spynet_grad_attrib.py)
        self.recompute_chains = bool(recompute_chains)  # the recurrent residual chains keep their inputs only and are re-run in the backward
        self.fp8_chains = bool(fp8_chains)  # conv1 / conv2 of the chains' residual blocks in fp8 (e4m3, block-scaled; compute_dtype bf16, 144 / 112 channels)
        self.spynet = SPyNet(spynet_pretrained) if spynet_pretrained is not None else None

        enc_dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths[:self.num_enc_layers]))]
        dec_dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths[self.num_enc_layers:]))][::-1]
        if not is_train:
            enc_dpr, dec_dpr = [0.] * len(enc_dpr), [0.] * len(dec_dpr)
        self.chunk_ratio = [float(Fraction(r)) for r in chunk_ratios]
        self.chunk_h = [int(self.init_H * x) for x in self.chunk_ratio]
        self.chunk_w = [int(self.init_W * x) for x in self.chunk_ratio]
        self.local_fuse = if_local_fuse
        if if_local_fuse:
            self.local_cnn = nn.Conv2d(embed_dim[0], embed_dim[0], 3, 1, 1)
        self.input_proj = InputProj(in_chans, embed_dim[0])

        def stage(dim, depth, heads, ch, cw, dpr, ws, nk, al, rn, th):
            return Mlp_encoder(dim, depth, heads, ch, cw, mlp_ratio, n_groups, qkv_bias, dpr, ws, nk, al, temporal_empty, rn, th,
                               flow_smooth, smooth_region_range, ffn_type, r_scaling, twins, traj_scale, m_scaling, if_local_fuse,
                               channel_mixer)

        self.encoder_layers, self.upsample, self.downsample = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for i in range(self.num_enc_layers):
            self.encoder_layers.append(stage(embed_dim[i], depths[i], num_heads[i], self.chunk_h[i], self.chunk_w[i],
                                             enc_dpr[sum(depths[:i]):sum(depths[:i + 1])], window_sizes[i], traj_keyframes_n[i],
                                             temporal_type[i], traj_res_n[i], traj_heads[i]))
            if i != self.num_enc_layers - 1:
                self.downsample.append(UpdownkeepSampling(embed_dim[i], embed_dim[i + 1], "down"))
            else:
                self.upsample.append(UpdownkeepSampling(embed_dim[i], embed_dim[i + 1], "up"))
        self.decoder_layers = nn.ModuleList()
        ne = self.num_enc_layers
        for i in range(self.num_dec_layers):
            self.decoder_layers.append(stage(embed_dim[i + ne], depths[i + ne], num_heads[i + ne], self.chunk_h[-i - 2],
                                             self.chunk_w[-i - 2], dec_dpr[sum(dec_depths[:i]):sum(dec_depths[:i + 1])],
                                             window_sizes[i + ne], traj_keyframes_n[-i - 2], temporal_type[-i - 2], traj_res_n[i + ne],
                                             traj_heads[-i - 2]))
            if i != self.num_dec_layers - 1:
                self.upsample.append(UpdownkeepSampling(embed_dim[i + ne], embed_dim[i + ne + 1], "up"))

        self.upconv1 = nn.Conv2d(embed_dim[-1], embed_dim[-1] * 4, 3, 1, 1)
        self.upconv2 = nn.Conv2d(embed_dim[-1], 64 * 4, 3, 1, 1)
        self.HRconv = nn.Conv2d(64, 64, 3, 1, 1)
        self.conv_last = nn.Conv2d(64, 3, 3, 1, 1)
        self.mdsc = mdsc
        if mdsc:
            self.sc_64_16 = nn.Sequential(nn.Conv2d(embed_dim[0], embed_dim[2], 1, 1, 0), nn.GroupNorm(1, embed_dim[2]), nn.ReLU())
            self.sc_32_8 = nn.Sequential(nn.Conv2d(embed_dim[1], embed_dim[3], 1, 1, 0), nn.GroupNorm(1, embed_dim[3]), nn.ReLU())
        self.mlp_wd_param = [p for name, p in self.named_parameters() if ".mlp_blocks." in name]
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # ---------------------------------------------------------------------------------------------- flows
    def check_frames_mirror(self, lrs):
        self.frames_mirror = False
        if lrs.size(1) % 2 == 0:
            a, b = torch.chunk(lrs, 2, dim=1)
            if torch.linalg.norm(a - b.flip(1)) == 0:
                self.frames_mirror = True

    def compute_flow(self, lrs):
        """Per encoder scale: SPyNet forward/backward flows (models/vmg.py:435-464)."""
        B, T, C, H, W = lrs.shape
        fwd, bwd = [], []
        for i in range(self.num_enc_layers):
            h, w = H // (2 ** i), W // (2 ** i)
            xi = F.adaptive_avg_pool2d(lrs.reshape(B * T, C, H, W), (h, w)).reshape(B, T, C, h, w)
            a = xi[:, :-1].reshape(-1, C, h, w)
            b = xi[:, 1:].reshape(-1, C, h, w)
            if self.frames_mirror:
                ff = self.spynet(b, a, self.spynet_dtype or self.compute_dtype).reshape(B, T - 1, 2, h, w)
                fb = ff.flip(1)
            else:  # both directions in ONE SPyNet pass (twice the batch, half the launches; per-sample results unchanged)
                both = self.spynet(torch.cat([b, a], 0), torch.cat([a, b], 0), self.spynet_dtype or self.compute_dtype).contiguous()
                ff, fb = FH.split_halves(both)
                ff, fb = ff.reshape(B, T - 1, 2, h, w), fb.reshape(B, T - 1, 2, h, w)
            fwd.append(ff)
            bwd.append(fb)
        return fwd, bwd

    # ---------------------------------------------------------------------------------------------- trunk
    def _mdsc(self, seq, x):
        """adaptive_max_pool2d /4 -> conv1x1 -> GroupNorm(1) -> ReLU (models/vmg.py:388-400, 519, 525), channels-last on the HIP kernels."""
        B, T, H, W, C = x.shape
        if H % 4 or W % 4:
            raise HipError("the multi-scale skip pools by 4: stage sizes must be multiples of 4 (they are: the input is padded to a multiple of 8)")
        f = FH.max_pool(x.reshape(B * T, H, W, C), 4)
        f = FH.conv2d([f], seq[0].weight, seq[0].bias, B * T, H // 4, W // 4, ks=1)
        f = FH.group_norm1_relu(f, seq[1].weight, seq[1].bias, seq[1].eps)
        return f.reshape(B, T, H // 4, W // 4, -1)

    def forward_features_multi_stages(self, x, ff, fb):
        enc, dec, down, up = self.encoder_layers, self.decoder_layers, self.downsample, self.upsample
        # (a stage output with several consumers -- next stage, long skip, multi-scale skip -- goes out as one handle per consumer: FH.fan_out)
        x1, x1_skip, x1_ms = FH.fan_out(enc[0](x, ff[0], fb[0]), 3)
        x1_3 = self._mdsc(self.sc_64_16, x1_ms) if self.mdsc else 0
        x2, x2_skip, x2_ms = FH.fan_out(enc[1](down[0](x1), ff[1], fb[1]), 3)
        x2_4 = self._mdsc(self.sc_32_8, x2_ms) if self.mdsc else 0
        x3, x3_skip = FH.fan_out(enc[2](down[1](x2), ff[2], fb[2]), 2)
        x4 = enc[3](down[2](x3 + x1_3), ff[3], fb[3])
        x5 = dec[0](up[0](x4 + x2_4), ff[2], fb[2])
        x6 = dec[1](up[1](x5 + x3_skip), ff[1], fb[1])
        x7 = dec[2](up[2](x6 + x2_skip), ff[0], fb[0])
        return x7 + x1_skip

    def forward_features_few_stages(self, x, ff, fb):
        x1, x1_skip = FH.fan_out(self.encoder_layers[0](x, ff[0], fb[0]), 2)
        x2 = self.encoder_layers[1](self.downsample[0](x1), ff[1], fb[1])
        x3 = self.decoder_layers[0](self.upsample[0](x2), ff[0], fb[0])
        return x3 + x1_skip

    def reconstruct(self, y, N, H, W):
        """The 4x head on channels-last features (N,H,W,C) -> (N,4H,4W,3): upconv1 / PixelShuffle / lrelu, upconv2 / PixelShuffle / lrelu,
        HRconv / lrelu, conv_last (models/vmg.py:629-632)."""
        o = conv(self.upconv1, [y], N, H, W, act=ACT_LRELU, slope=0.1, pixel_shuffle=True)
        o = conv(self.upconv2, [o], N, 2 * H, 2 * W, act=ACT_LRELU, slope=0.1, pixel_shuffle=True)
        o = conv(self.HRconv, [o], N, 4 * H, 4 * W, act=ACT_LRELU, slope=0.1)
        return conv(self.conv_last, [o], N, 4 * H, 4 * W, fuse_src_act=True)  # (HRconv's leaky-ReLU derivative in conv_last's data-gradient epilogue: one HR-size pass less)

    def forward(self, x, flow_pretrained=None, config_amp=None):
        B, D, C, H, W = x.size()
        assert H >= 64 and W >= 64, "The height and width must larger than 64."
        hip.require_cuda(x)
        if self.spynet is None:
            raise HipError("VMG.spynet is None: the trajectory modules need optical flow (the reference crashes here too, "
                           "models/trajectory.py:329); construct with spynet_pretrained or attach SPyNet(None)")
        if getattr(self, "_recompute_applied", None) != (self.recompute_chains, self.fp8_chains):
            for m in self.modules():
                if isinstance(m, ResidualBlocksWithInputConv):
                    m.recompute, m.fp8 = self.recompute_chains, self.fp8_chains
            self._recompute_applied = (self.recompute_chains, self.fp8_chains)
        FH.DROP.begin(x.device, self.training)  # the DropPath masks of this pass in one draw (functional._DropPlan)
        # T1 (SURVEY trap): every MorphFC mixer multiplies its mlp_h / mlp_w weights by Gamma at each forward call -- all of them here, in one
        # launch, with their packs rebuilt in one more; each mixer then skips its own multiply for this call
        mixers = self.__dict__.get("_morph_mixers")
        if mixers is None:
            mixers = self.__dict__["_morph_mixers"] = [m for m in self.modules() if isinstance(m, Enhanced_MorphFCs_decay)]
        if mixers:
            FH.decay_weights_and_repack([m.mlp_h[0].weight for m in mixers] + [m.mlp_w[0].weight for m in mixers],
                                        [m.gamma_h for m in mixers] + [m.gamma_w for m in mixers])
            for m in mixers:
                m._t1_done = True
        if torch.is_grad_enabled():
            FH.DEFERRED.begin_forward()  # per-pass use counts of the deferred weight gradients (functional._DeferredWgrad)
        # the kernels take fp32 / the module's compute dtype; an enclosing torch.autocast (tools/Trainer.py:132-143) must not
        # re-type the few torch ops left in here
        with torch.autocast("cuda", enabled=False):
            return self._forward(x)

    def _forward(self, x):
        B, D, C, H, W = x.size()
        in_dtype = x.dtype
        x = x.float()
        if self.__dict__.get("_mirror_known") is None:
            self.check_frames_mirror(lrs=x)  # (a host decision on device data: infer.GraphedModel takes it before the replay and passes it in)
        else:
            self.frames_mirror = bool(self._mirror_known)
        up = F.interpolate(x.reshape(B * D, C, H, W), scale_factor=4, mode="bilinear", align_corners=False)  # == trilinear with D kept
        Hp = int(np.ceil(H / self.scale)) * self.scale
        Wp = int(np.ceil(W / self.scale)) * self.scale
        x = F.pad(x, (0, Wp - W, 0, Hp - H, 0, 0), mode="replicate")
        ff, fb = self.compute_flow(x)
        cd = self.compute_dtype
        xin = x.reshape(B * D, C, Hp, Wp).permute(0, 2, 3, 1).contiguous().to(cd)  # channels-last frames
        feat = conv(self.input_proj.proj[0], [xin], B * D, Hp, Wp, act=ACT_LRELU, slope=0.01).reshape(B, D, Hp, Wp, -1)
        if self.num_layers > 3:
            y = self.forward_features_multi_stages(feat, ff, fb)
        else:
            y = self.forward_features_few_stages(feat, ff, fb)
        N = B * D
        if self.local_fuse:
            y = conv(self.local_cnn, [y], N, Hp, Wp, res=feat.reshape(N, Hp, Wp, -1))
        y = y.reshape(N, Hp, Wp, -1)
        if (Hp, Wp) != (H, W):
            y = y[:, :H, :W].contiguous()
        o = self.reconstruct(y, N, H, W)
        out = o.float().permute(0, 3, 1, 2) + up
        return out.reshape(B, self.num_out_frames, -1, 4 * H, 4 * W).to(in_dtype)
