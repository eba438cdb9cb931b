"""Shared helpers for the parity tests (test infrastructure)."""
import math

import numpy as np
import torch


def build_product(cfg, compute_dtype=torch.float32, device="cuda"):
    """vmg_amd.VMG from an oracle VMGConfig (the same keyword mapping oracle/gen_golden.py uses for the reference)."""
    import vmg_amd
    ne = cfg.num_enc_layers
    m = vmg_amd.VMG(embed_dim=list(cfg.embed_dim), depths=list(cfg.depths), num_heads=list(cfg.num_heads), num_frames=cfg.num_frames,
                    window_sizes=[list(w) for w in cfg.window_sizes], mdsc=cfg.mdsc, if_concat=False, mlp_ratio=cfg.mlp_ratio,
                    n_groups=cfg.n_groups, spynet_pretrained=None, image_size=list(cfg.image_size), is_train=cfg.is_train,
                    traj_win=list(cfg.traj_win), traj_keyframes_n=list(cfg.traj_keyframes_n), traj_heads=list(cfg.traj_heads),
                    temporal_type=list(cfg.temporal_type), temporal_empty=cfg.temporal_empty, traj_res_n=list(cfg.traj_res_n),
                    spatial_type=list(cfg.spatial_type), flow_smooth=cfg.flow_smooth, smooth_region_range=cfg.smooth_region_range,
                    symm_act="tanh", ffn_type=cfg.ffn_type, mixer_type=["mlps"] * ne, mixer_n=[None] * ne, r_scaling=cfg.r_scaling,
                    chunk_ratios=list(cfg.chunk_ratios), twins=list(cfg.twins), traj_scale=cfg.traj_scale, m_scaling=cfg.m_scaling,
                    if_local_fuse=cfg.if_local_fuse, channel_mixer=cfg.channel_mixer, compute_dtype=compute_dtype)
    m.spynet = vmg_amd.SPyNet(None)
    return m.to(device) if device else m


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    """PSNR of two [0,1] image tensors after the reference's clamp/x255/round (tools/Tester.py:249-250, utils/metrics.py:11-26)."""
    qa = (a.clamp(0, 1) * 255).round().double()
    qb = (b.clamp(0, 1) * 255).round().double()
    mse = float(((qa - qb) ** 2).mean())
    return float("inf") if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse))
