"""create_model(config) -> nn.Module: the reference's factory (models/__init__.py:6-49) for the HIP-backed VMG.

Missing config keys behave as VMG.__init__ defaults (SURVEY traps T4/T5: the shipped VMG-REDS.yml lacks keys that
the reference's factory indexes, and `if_print` is in no config)."""
from __future__ import annotations

import torch

from .model import VMG

_KEYMAP = {  # yaml key -> VMG keyword
    "embed_dim": "embed_dim", "depths": "depths", "mlp_ratio": "mlp_ratio", "n_groups": "n_groups", "num_heads": "num_heads",
    "window_sizes": "window_sizes", "num_frames": "num_frames", "back_RBs": "back_RBs", "spynet": "spynet_pretrained",
    "if_print": "if_print", "ltam": "ltam", "traj_win": "traj_win", "traj_keyframes_n": "traj_keyframes_n", "traj_heads": "traj_heads",
    "temporal_type": "temporal_type", "temporal_empty": "temporal_empty", "traj_res_n": "traj_res_n", "deform_groups": "deform_groups",
    "max_res_scale": "max_residual_scale", "spatial_type": "spatial_type", "use_mdsc": "mdsc", "if_concat": "if_concat",
    "flow_smooth": "flow_smooth", "smooth_region_range": "smooth_region_range", "ret_decay": "retention_decay",
    "non_linear": "non_linear", "gating": "gating", "if_symm": "symm", "symm_act": "symm_act", "relu_scale": "relu_scale",
    "relu_scale_norm": "relu_scale_norm", "ffn_type": "ffn_type", "mixer_type": "mixer_type", "mixer_n": "mixer_n",
    "r_scaling": "r_scaling", "chunk_ratios": "chunk_ratios", "traj_mode": "traj_mode", "twins": "twins", "traj_scale": "traj_scale",
    "traj_refine": "traj_refine", "m_scaling": "m_scaling", "if_local_fuse": "if_local_fuse", "channel_mixer": "channel_mixer",
}


def create_model(config):
    net = config["network"]
    if config["model"] != "VMG":
        raise NotImplementedError("Model [{:s}] not recognized.".format(str(config["model"])))
    shape_r = config["dataset"]["image_shape_r"]
    image_size = [int(shape_r[1] / config["scale"]), int(shape_r[2] / config["scale"])]
    kw = {}
    for yk, vk in _KEYMAP.items():
        try:
            v = net[yk]
        except KeyError:
            continue
        if v is None and yk not in ("spynet", "traj_refine"):
            continue  # NoneDict gives None for absent keys: fall back to the constructor default
        kw[vk] = v
    cd = config.get("compute_dtype", "float32") if hasattr(config, "get") else "float32"
    kw["compute_dtype"] = {"float32": torch.float32, "bfloat16": torch.bfloat16, None: torch.float32}[cd]
    kw["recompute_chains"] = bool(config.get("recompute_chains", False)) if hasattr(config, "get") else False  # optional: activation recompute
    return VMG(image_size=image_size, is_train=config["is_train"], **kw)
