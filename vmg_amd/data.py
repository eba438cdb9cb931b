"""Synthetic REDS-shaped clips for benchmarks and smoke runs: there is no dataset in the container."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def synthetic_clip(B: int, T: int, H: int, W: int, seed: int = 1234, device="cpu") -> torch.Tensor:
    """LR clip (B,T,3,H,W) in [0,1]: 3x3 box-filtered noise shifted 1 px per frame along x (coherent motion for SPyNet)."""
    g = torch.Generator().manual_seed(seed)
    base = F.avg_pool2d(torch.rand(B, 3, H + 2, W + T + 2, generator=g), 3, 1)
    return torch.stack([base[..., t:t + W] for t in range(T)], 1).contiguous().to(device)


def synthetic_target(x: torch.Tensor, seed: int = 4321) -> torch.Tensor:
    """HR target: bicubic x4 of the clip + N(0, 0.01^2) noise."""
    B, T, C, H, W = x.shape
    g = torch.Generator().manual_seed(seed)
    up = F.interpolate(x.reshape(B * T, C, H, W).float().cpu(), scale_factor=4, mode="bicubic", align_corners=False)
    up = up + 0.01 * torch.randn(up.shape, generator=g)
    return up.reshape(B, T, C, 4 * H, 4 * W).to(x.device)


REDS_FEW_LEVELS = dict(  # network block of the reference's configs/VMG-REDS-few_levels.yml
    embed_dim=[144, 144, 144], depths=[4, 4, 4], num_heads=[4, 8, 4], mlp_ratio=2, n_groups=1,
    window_sizes=[[2, 8, 8], [4, 8, 8], [2, 8, 8]], back_RBs=0, ltam=True, traj_win=[16, None], traj_keyframes_n=[3, None],
    traj_heads=[4, None], temporal_type=[False, None], temporal_empty=True, traj_res_n=[15, 0, 15], deform_groups=[8, 16, 8],
    max_residual_scale=[1, 2, 1], spatial_type=[False, False], mdsc=False, if_concat=False, flow_smooth=True, smooth_region_range=4,
    retention_decay=True, non_linear=True, gating=True, symm=True, symm_act="tanh", relu_scale=True, relu_scale_norm=False,
    ffn_type="ffn_cnn", mixer_type=["mlps", "mlps"], mixer_n=[None, None], r_scaling=0.1, chunk_ratios=["1/8", "1/4"],
    traj_mode="wins", twins=[2, 2], traj_scale=True, traj_refine=None, m_scaling=1.0, if_local_fuse=True, channel_mixer="rcab")

REDS_FULL = dict(  # network block of the reference's configs/VMG-REDS.yml (4 encoder / 3 decoder levels); keys that file lacks keep VMG.__init__'s defaults
    embed_dim=[112, 224, 224, 448, 224, 224, 112], depths=[4, 4, 2, 2, 2, 4, 4], num_heads=[4, 8, 8, 16, 8, 8, 4], mlp_ratio=6, n_groups=4,
    window_sizes=[[2, 8, 8], [4, 8, 8], [6, 8, 8], [8, 8, 8], [6, 8, 8], [4, 8, 8], [2, 8, 8]], back_RBs=0, ltam=True,
    traj_win=[16, None, None, None], traj_keyframes_n=[3, None, None, None], traj_heads=[4, None, None, None],
    temporal_type=[False, None, None, None], temporal_empty=True, traj_res_n=[15, 0, 0, 0, 0, 0, 15], spatial_type=[False] * 4, mdsc=True,
    if_concat=False, flow_smooth=True, smooth_region_range=4, retention_decay=True, non_linear=True, gating=True, symm=True, symm_act="tanh",
    relu_scale=True, relu_scale_norm=False, ffn_type="ffn_cnn", mixer_type=["mlps"] * 4, mixer_n=[None] * 4, r_scaling=0.1,
    chunk_ratios=["1/8", "1/4", "3/16", "1/8"], traj_mode="wins", twins=[2, 2], traj_scale=True, traj_refine=None, m_scaling=1.0)
