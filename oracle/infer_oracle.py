"""CPU restatement of the reference's sliding-window inference (tools/Tester.py) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  Pinned by the fixtures
tests/golden/infer_*.npz, which oracle/gen_golden.py produced by calling the unmodified Tester.test_image / test_clips /
test_clips_max (through a stub `self`) in the build container.

The windows are visited in the reference's order: the model is stateful, so the order is part of the result.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np
import torch


def tile_starts(total: int, size: int, overlap: int) -> List[int]:
    """tools/Tester.py:113-114, :151-152: starts every (size - overlap), plus one window flush with the end."""
    stride = size - overlap
    return list(range(0, total - size, stride)) + [max(0, total - size)]


def test_image(model: Callable, inputs: torch.Tensor, test_spatial: Sequence[int], overlap: int, scale: int) -> torch.Tensor:
    """tools/Tester.py:107-141.  Spatial tiles; the half-overlap borders of a tile that has a neighbour on that side are
    zeroed in the output AND in the weight map -- in HIGH-resolution pixels but with the LOW-resolution count overlap//2
    (the slices are `-overlap//2:` = the last ceil(overlap/2) rows and `:overlap//2` = the first floor(overlap/2))."""
    B, T, C, H, W = inputs.shape
    th, tw = test_spatial
    hs, ws = tile_starts(H, th, overlap), tile_starts(W, tw, overlap)
    E = inputs.new_zeros(B, T, C, H * scale, W * scale)
    Wt = torch.zeros_like(E)
    for h in hs:
        for w in ws:
            out = model(inputs[..., h:h + th, w:w + tw])
            mask = torch.ones_like(out)
            if h < hs[-1]:
                out[..., -overlap // 2:, :] *= 0
                mask[..., -overlap // 2:, :] *= 0
            if w < ws[-1]:
                out[..., :, -overlap // 2:] *= 0
                mask[..., :, -overlap // 2:] *= 0
            if h > hs[0]:
                out[..., :overlap // 2, :] *= 0
                mask[..., :overlap // 2, :] *= 0
            if w > ws[0]:
                out[..., :, :overlap // 2] *= 0
                mask[..., :, :overlap // 2] *= 0
            E[..., h * scale:(h + th) * scale, w * scale:(w + tw) * scale].add_(out)
            Wt[..., h * scale:(h + th) * scale, w * scale:(w + tw) * scale].add_(mask)
    return E.div_(Wt)


def test_clips(model: Callable, inputs: torch.Tensor, num_frames: int, overlap_frames: int, test_spatial: Optional[Sequence[int]] = None,
               overlap_spatial: Optional[int] = None, scale: int = 4) -> torch.Tensor:
    """tools/Tester.py:143-175.  Temporal windows; frames in the half-overlap towards a neighbouring window are dropped."""
    B, T, C, H, W = inputs.shape
    E = inputs.new_zeros(B, T, C, H * scale, W * scale)
    N = inputs.new_zeros(B, T, 1, 1, 1)
    ts = tile_starts(T, num_frames, overlap_frames)
    for t in ts:
        clip = inputs[:, t:t + num_frames]
        out = model(clip) if overlap_spatial is None else test_image(model, clip, test_spatial, overlap_spatial, scale)
        n = inputs.new_ones(B, num_frames, 1, 1, 1)
        if overlap_frames > 0:
            if t < ts[-1]:
                out[:, -overlap_frames // 2:] *= 0
                n[:, -overlap_frames // 2:] *= 0
            if t > ts[0]:
                out[:, :overlap_frames // 2] *= 0
                n[:, :overlap_frames // 2] *= 0
        E[:, t:t + num_frames].add_(out)
        N[:, t:t + num_frames].add_(n)
    return E.div_(N)


def psnr_float(image_test: np.ndarray, image_true: np.ndarray) -> float:
    """skimage.metrics.peak_signal_noise_ratio(image_test, image_true) as tools/Tester.py:208 calls it (first argument is
    skimage's `image_true`): float images -> data_range 1 when that image has no negative value, else 2; float64 mean."""
    a, b = np.asarray(image_test), np.asarray(image_true)
    data_range = 1.0 if a.min() >= 0 else 2.0
    err = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2, dtype=np.float64)
    with np.errstate(divide="ignore"):
        return float(10 * np.log10((data_range ** 2) / err))


def psnr_exceed_check(psnr: float) -> float:
    """tools/Tester.py:24-34: an infinite PSNR is replaced by 10*log10(255^2 / 0.65025)."""
    if psnr >= float("inf"):
        return float(10 * np.log10(255.0 ** 2 / 0.65025))
    if psnr < 0:
        raise Exception("Wrong way of calculating psnr.")
    return psnr


def test_clips_max(model: Callable, inputs: torch.Tensor, HR: torch.Tensor, num_frames: int, overlap_frames: int,
                   test_spatial: Optional[Sequence[int]] = None, overlap_spatial: Optional[int] = None, scale: int = 4) -> torch.Tensor:
    """tools/Tester.py:178-216 (REDS): every frame takes the window whose output has the highest PSNR against HR
    (frames a window does not cover score 0 there; batch 1, like the reference's .squeeze())."""
    B, T, C, H, W = inputs.shape
    ts = tile_starts(T, num_frames, overlap_frames)
    E = inputs.new_zeros(B, T, len(ts), C, H * scale, W * scale)
    psnrs = inputs.new_zeros(B, T, len(ts))
    for idx, t in enumerate(ts):
        clip = inputs[:, t:t + num_frames]
        hr = HR[:, t:t + num_frames]
        out = model(clip) if overlap_spatial is None else test_image(model, clip, test_spatial, overlap_spatial, scale)
        for i in range(num_frames):
            a = out[:, i].squeeze().permute(1, 2, 0).contiguous().cpu().clamp(0, 1).numpy()
            b = hr[:, i].squeeze().permute(1, 2, 0).contiguous().cpu().clamp(0, 1).numpy()
            psnrs[:, t + i, idx] = psnr_exceed_check(psnr_float(a, b))
        E[:, t:t + num_frames, idx].add_(out)
    _, max_idx = torch.max(psnrs, dim=-1)
    max_idx = max_idx[:, :, None, None, None, None].expand(-1, -1, -1, C, H * scale, W * scale)
    return torch.gather(E, dim=2, index=max_idx).squeeze()  # (T, C, 4H, 4W) for the batch of 1 the reference runs


def to_uint8(outputs: torch.Tensor) -> np.ndarray:
    """tools/Tester.py:249-250: clamp to [0,1], *255 in float32, numpy round (half to even), uint8; (T,H,W,C) layout."""
    o = outputs.cpu().squeeze().clamp(0, 1).numpy()
    return np.round(np.ascontiguousarray(o.squeeze().transpose(0, 2, 3, 1)) * 255.0).astype(np.uint8)


def fake_sr_model(scale: int = 4):
    """A deterministic stand-in network for the harness fixtures (ours, not the reference's): nearest x`scale` of the input
    plus a ramp in TILE-LOCAL coordinates and a call counter, so overlapping tiles disagree and the visiting order shows."""
    state = {"calls": 0}

    def model(x: torch.Tensor) -> torch.Tensor:
        B, T, C, h, w = x.shape
        up = x.repeat_interleave(scale, -2).repeat_interleave(scale, -1)
        ii = torch.arange(h * scale, dtype=x.dtype, device=x.device)[:, None] / (h * scale)
        jj = torch.arange(w * scale, dtype=x.dtype, device=x.device)[None, :] / (w * scale)
        tt = torch.arange(T, dtype=x.dtype, device=x.device)[None, :, None, None, None]
        state["calls"] += 1
        return 0.5 * up + 0.2 * (ii + 0.5 * jj) + 0.03 * tt + 0.001 * state["calls"]

    return model
