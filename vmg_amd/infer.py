"""Sliding-window inference for the VMG hot path: counterpart of the reference's tools/Tester.py:107-251.

Same window lists, same visiting order (the network is stateful, SURVEY T1, so the order is part of the result), same
half-overlap conventions (`-overlap//2:` drops ceil(overlap/2) trailing rows/frames, `:overlap//2` floor(overlap/2) leading
ones, counted in OUTPUT pixels with the LOW-resolution overlap, exactly as the reference slices them).  The canvases stay
in HBM; each tile is folded in by one HIP kernel (vmg_tile_accumulate) and the division / clamp / uint8 rounding is one
more (vmg_tile_finalize).  There is no CPU path: tensors must live on the GPU.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from . import hip
from .hip import HipError


class GraphedModel:
    """model(x) replayed from a captured hipGraph, one graph per input shape (the sliding-window harness calls the network 18 times per sequence on
    (1, 50, 3, 128, 128) tiles: ~7 000 launches per call issued from Python otherwise).  The network is STATEFUL (every call multiplies
    the MorphFC mixer weights by Gamma in place) and the capture procedure needs warm-up calls: the mixer weights are saved before and put back
    after them, so the first replay is call #1 exactly as without the wrapper; the decay itself is part of the graph (it runs at every replay).
    Results are the same bits as the eager calls (the same kernels on the same data; tests/test_infer_gpu.py)."""

    def __init__(self, model: torch.nn.Module, warmup: int = 3):
        self.model, self.warmup, self.graphs = model, int(warmup), {}

    def _mixer_weights(self):
        return [p for n, p in self.model.named_parameters() if n.endswith("mlp_h.0.weight") or n.endswith("mlp_w.0.weight")]

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        hip.require_cuda(x)
        # the mirrored-clip test (models/vmg.py:426-432) is a host decision on device data: taken here, once per call, and part of the graph's key
        self.model.check_frames_mirror(lrs=x.float())
        mirror = bool(self.model.frames_mirror)
        self.model._mirror_known = mirror
        try:
            return self._call(x, mirror)
        finally:
            self.model._mirror_known = None

    def _call(self, x: torch.Tensor, mirror: bool) -> torch.Tensor:
        key = (tuple(x.shape), x.dtype, mirror)
        ent = self.graphs.get(key)
        if ent is None:
            from . import functional as FH
            static_in = x.clone()
            ws = self._mixer_weights()
            saved = [w.detach().clone() for w in ws]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for i in range(max(self.warmup, 6)):  # until the cached weight packs / repack plans have settled (see train.TrainStep.capture)
                    stamp = FH._PACK_STAMP[0]
                    self.model(static_in)
                    if i + 1 >= self.warmup and FH._PACK_STAMP[0] == stamp and FH._VOL_STATE["stamp"] == stamp:
                        break
                for w, s0 in zip(ws, saved):
                    w.copy_(s0)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                static_out = self.model(static_in)
            ent = self.graphs[key] = (g, static_in, static_out)
        g, static_in, static_out = ent
        static_in.copy_(x)
        g.replay()
        return static_out.clone()


def tile_starts(total: int, size: int, overlap: int) -> List[int]:
    """Window starts of tools/Tester.py:113-114 / :151-152: every (size - overlap), plus one window flush with the end."""
    stride = size - overlap
    if stride <= 0:
        raise ValueError(f"overlap {overlap} must be smaller than the window {size}")
    return list(range(0, total - size, stride)) + [max(0, total - size)]


def _accumulate(patch: torch.Tensor, E: torch.Tensor, Wt: torch.Tensor, oh: int, ow: int, margins) -> None:
    hip.require_cuda(patch, E, Wt)
    if E.dtype != torch.float32 or Wt.dtype != torch.float32 or not E.is_contiguous() or not Wt.is_contiguous():
        raise HipError("canvases must be contiguous fp32")
    p = patch.contiguous()
    ph, pw = p.shape[-2:]
    planes = p.numel() // (ph * pw)
    EH, EW = E.shape[-2:]
    if E.numel() // (EH * EW) != planes:
        raise HipError("tile and canvas disagree on the number of planes")
    top, bottom, left, right = margins
    hip.check(hip.lib().vmg_tile_accumulate(hip.dtype_code(p.dtype), p.data_ptr(), E.data_ptr(), Wt.data_ptr(), planes, ph, pw, EH, EW, oh, ow,
                                            top, bottom, left, right, hip.stream_ptr()), "vmg_tile_accumulate")


def _finalize(E: torch.Tensor, Wt: torch.Tensor, want_u8: bool = False):
    out = None if want_u8 else torch.empty_like(E)
    u8 = torch.empty(E.shape, dtype=torch.uint8, device=E.device) if want_u8 else None
    hip.check(hip.lib().vmg_tile_finalize(E.data_ptr(), Wt.data_ptr(), out.data_ptr() if out is not None else None,
                                          u8.data_ptr() if u8 is not None else None, E.numel(), hip.stream_ptr()), "vmg_tile_finalize")
    return u8 if want_u8 else out


@torch.no_grad()
def test_image(model: Callable, inputs: torch.Tensor, test_spatial: Sequence[int], overlap: int, scale: int = 4) -> torch.Tensor:
    """tools/Tester.py:107-141: spatial tiles of `test_spatial` with `overlap` LR pixels between neighbours."""
    hip.require_cuda(inputs)
    B, T, C, H, W = inputs.shape
    th, tw = test_spatial
    hs, ws = tile_starts(H, th, overlap), tile_starts(W, tw, overlap)
    E = torch.zeros(B, T, C, H * scale, W * scale, dtype=torch.float32, device=inputs.device)
    Wt = torch.zeros_like(E)
    lead = overlap // 2  # `:overlap//2` of the reference
    for h in hs:
        for w in ws:
            out = model(inputs[..., h:h + th, w:w + tw])
            # `-overlap//2:` drops the last ceil(overlap/2) rows -- and, faithfully, EVERY row when overlap == 0 (the slice is
            # then `0:`; the reference yields 0/0 = NaN there: tiles without overlap are not a supported setting of it)
            trail_h = -(-overlap // 2) if overlap > 0 else out.shape[-2]
            trail_w = -(-overlap // 2) if overlap > 0 else out.shape[-1]
            margins = (lead if h > hs[0] else 0, trail_h if h < hs[-1] else 0, lead if w > ws[0] else 0, trail_w if w < ws[-1] else 0)
            _accumulate(out, E, Wt, h * scale, w * scale, margins)
    return _finalize(E, Wt).to(inputs.dtype)


@torch.no_grad()
def test_clips(model: Callable, inputs: torch.Tensor, num_frames: int, overlap_frames: int, test_spatial: Optional[Sequence[int]] = None,
               overlap_spatial: Optional[int] = None, scale: int = 4) -> torch.Tensor:
    """tools/Tester.py:143-175: temporal windows of `num_frames` with `overlap_frames` shared frames."""
    hip.require_cuda(inputs)
    B, T, C, H, W = inputs.shape
    E = torch.zeros(B, T, C, H * scale, W * scale, dtype=torch.float32, device=inputs.device)
    N = torch.zeros(B, T, 1, 1, 1, dtype=torch.float32, device=inputs.device)
    ts = tile_starts(T, num_frames, overlap_frames)
    lead, trail = overlap_frames // 2, -(-overlap_frames // 2)
    for t in ts:
        clip = inputs[:, t:t + num_frames]
        out = model(clip) if overlap_spatial is None else test_image(model, clip, test_spatial, overlap_spatial, scale)
        lo = lead if (overlap_frames > 0 and t > ts[0]) else 0
        hi = num_frames - (trail if (overlap_frames > 0 and t < ts[-1]) else 0)
        # frames are whole planes: a slice add is already one pass
        E[:, t + lo:t + hi].add_(out[:, lo:hi].float())
        N[:, t + lo:t + hi].add_(1.0)
    return E.div_(N).to(inputs.dtype)


def _psnr01(a: torch.Tensor, b: torch.Tensor) -> float:
    """skimage.metrics.peak_signal_noise_ratio on [0,1]-clamped float images (data_range 1), float64 mean, with the
    reference's replacement of an infinite value (tools/Tester.py:24-34, :204-210)."""
    err = float(((a.clamp(0, 1).double() - b.clamp(0, 1).double()) ** 2).mean())
    if err == 0.0:
        return float(10 * np.log10(255.0 ** 2 / 0.65025))
    v = 10.0 * math.log10(1.0 / err)
    if v < 0:
        raise Exception("Wrong way of calculating psnr.")
    return v


@torch.no_grad()
def test_clips_max(model: Callable, inputs: torch.Tensor, HR: torch.Tensor, num_frames: int, overlap_frames: int,
                   test_spatial: Optional[Sequence[int]] = None, overlap_spatial: Optional[int] = None, scale: int = 4) -> torch.Tensor:
    """tools/Tester.py:178-216 (REDS): per frame, the window whose output scores the highest PSNR against HR.  Returns
    (T, C, 4H, 4W) like the reference's .squeeze() for its batch of one."""
    hip.require_cuda(inputs, HR)
    B, T, C, H, W = inputs.shape
    ts = tile_starts(T, num_frames, overlap_frames)
    E = torch.zeros(B, T, len(ts), C, H * scale, W * scale, dtype=torch.float32, device=inputs.device)
    psnrs = torch.zeros(B, T, len(ts), dtype=torch.float32)
    for idx, t in enumerate(ts):
        clip = inputs[:, t:t + num_frames]
        out = (model(clip) if overlap_spatial is None else test_image(model, clip, test_spatial, overlap_spatial, scale)).float()
        for i in range(num_frames):
            psnrs[:, t + i, idx] = _psnr01(out[:, i], HR[:, t + i].float())
        E[:, t:t + num_frames, idx].add_(out)
    _, max_idx = torch.max(psnrs, dim=-1)
    max_idx = max_idx.to(inputs.device)[:, :, None, None, None, None].expand(-1, -1, -1, C, H * scale, W * scale)
    return torch.gather(E, dim=2, index=max_idx).squeeze().to(inputs.dtype)


@torch.no_grad()
def to_uint8(outputs: torch.Tensor) -> np.ndarray:
    """tools/Tester.py:249-250: clamp, *255, round half to even, uint8, (T, H, W, C) on the host."""
    hip.require_cuda(outputs)
    o = outputs.float().squeeze().contiguous()
    ones = torch.ones_like(o)
    u8 = _finalize(o, ones, want_u8=True)
    return np.ascontiguousarray(u8.cpu().numpy().transpose(0, 2, 3, 1))
