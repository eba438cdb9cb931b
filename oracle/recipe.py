"""Deterministic weights-by-recipe and seeded synthetic inputs.  TEST INFRASTRUCTURE ONLY (see vmg_oracle.py).

No weights are committed: every tensor of a reference-format state dict is regenerated from its key name,
so the reference (in the build container), the oracle and the HIP module (on the GPU box) all see identical
parameters.  Buffers that the reference derives at construction (gamma_h/gamma_w, decay_v,
relative_position_index, spynet.mean/std) are rebuilt from their closed forms; oracle/gen_golden.py checks
those closed forms against the reference's own buffers before writing fixtures.
"""
from __future__ import annotations

import zlib
from typing import Dict, Sequence, Tuple

import torch

from . import vmg_oracle as O


def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def recipe_tensor(key: str, shape: Sequence[int], seed: int = 0) -> torch.Tensor:
    """One parameter tensor from its key.  sigma by kind: weights randn/sqrt(fan_in); biases 0.1 randn;
    LayerNorm/GroupNorm weights 1 + 0.1 randn; position tables 0.2 randn."""
    shape = tuple(shape)
    g = _gen(key, seed)
    leaf = key.split(".")[-1]
    r = torch.randn(shape, generator=g)
    if leaf in ("relative_pos_encoding", "relative_position_bias_table"):
        return 0.2 * r
    if leaf == "bias":
        return 0.1 * r
    if leaf == "weight" and len(shape) == 1:
        return 1.0 + 0.1 * r
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return r / float(fan_in) ** 0.5


def buffer_tensor(key: str, shape: Sequence[int], all_shapes: Dict[str, Tuple[int, ...]], ctx: dict) -> torch.Tensor:
    """Construction-time buffers of the reference, from their closed forms."""
    leaf = key.split(".")[-1]
    if leaf in ("gamma_h", "gamma_w"):
        chunk = ctx["chunk_of"](key)
        return O.decay_gamma(chunk, shape[0])
    if leaf == "decay_v":
        h = shape[0]
        return 1 - 2 ** (-5 - torch.arange(h - 1, -1, -1, dtype=torch.float32))  # models/trajectory.py:530
    if leaf == "relative_position_index":
        return O.relative_position_index(ctx["window_of"](key))
    if key == "spynet.mean":
        return torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    if key == "spynet.std":
        return torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    raise KeyError(key)


BUFFER_LEAVES = ("gamma_h", "gamma_w", "decay_v", "relative_position_index")


def is_buffer(key: str) -> bool:
    return key.split(".")[-1] in BUFFER_LEAVES or key in ("spynet.mean", "spynet.std")


def recipe_state_dict(shapes: Dict[str, Sequence[int]], seed: int = 0, chunk_of=None, window_of=None) -> Dict[str, torch.Tensor]:
    """Full reference-format state dict from {key: shape}."""
    ctx = {"chunk_of": chunk_of, "window_of": window_of}
    sd = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        sd[k] = buffer_tensor(k, shp, shapes, ctx) if is_buffer(k) else recipe_tensor(k, shp, seed)
    return sd


def vmg_chunk_lookup(cfg: O.VMGConfig):
    """Maps a '...spatial_mixing.gamma_h' key of a whole-VMG state dict to its chunk size."""
    ne = cfg.num_enc_layers

    def chunk_of(key: str) -> int:
        parts = key.split(".")
        i = int(parts[1])
        hw = cfg.chunk_h if parts[-1] == "gamma_h" else cfg.chunk_w
        return hw[i] if parts[0] == "encoder_layers" else hw[-i - 2]

    def window_of(key: str):
        parts = key.split(".")
        i = int(parts[1])
        return tuple(cfg.window_sizes[i] if parts[0] == "encoder_layers" else cfg.window_sizes[ne + i])

    return chunk_of, window_of


def synthetic_clip(B: int, T: int, H: int, W: int, seed: int = 1234) -> torch.Tensor:
    """LR clip in [0,1] with coherent motion: box-filtered noise shifted 1 px per frame along x."""
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(B, 3, H + 2, W + T + 2, generator=g)
    base = torch.nn.functional.avg_pool2d(base, 3, 1)  # (B,3,H,W+T)
    return torch.stack([base[..., t:t + W] for t in range(T)], 1).contiguous()


def synthetic_target(x: torch.Tensor, seed: int = 4321) -> torch.Tensor:
    """HR target = bicubic x4 of the clip + N(0, 0.01^2) noise."""
    B, T, C, H, W = x.shape
    g = torch.Generator().manual_seed(seed)
    up = torch.nn.functional.interpolate(x.reshape(B * T, C, H, W), scale_factor=4, mode="bicubic", align_corners=False)
    up = up + 0.01 * torch.randn(up.shape, generator=g)
    return up.reshape(B, T, C, 4 * H, 4 * W)


def seeded(shape, seed: int, scale: float = 1.0) -> torch.Tensor:
    return scale * torch.randn(tuple(shape), generator=torch.Generator().manual_seed(seed))
