"""Tensor-level wrappers over the C-ABI (no autograd here; see functional.py).

Every function takes/returns torch device tensors in channels-last layout and enqueues on torch's current
stream.  Shapes are validated on the host before a kernel is launched.
"""
from __future__ import annotations

import ctypes
import os
import struct
from typing import Optional, Sequence, Tuple

import torch

from . import hip
from .hip import ConvDesc, HipError


def _intarr(v: Sequence[int]):
    return (ctypes.c_int * len(v))(*[int(x) for x in v])


def cout_tiles_for(cout: int, dtype: torch.dtype = torch.bfloat16, ks: int = 3) -> int:
    """16-channel tiles per workgroup: 9 covers C = 144/288/576 exactly, 7 covers 112/224/448, 8 covers 128/256; otherwise
    the candidate with the least padding.  fp32 (the parity path) has twice the bytes per stage: at most 5 tiles."""
    t = (cout + 15) // 16
    if t <= 1:
        return 1
    if ks == 7:  # SPyNet's 7x7 convs (2 .. 64 output channels): 1, 2 or 4 tiles
        return 2 if t == 2 else 4
    if t <= 4:
        return 4
    best, waste = None, None
    for cand in ((5, 4) if dtype == torch.float32 else (9, 8, 7, 5)):
        w = (t + cand - 1) // cand * cand - t
        if waste is None or w < waste:
            best, waste = cand, w
    return best


class PackedConv:
    """Packed (device) weights of one convolution / linear for one direction (forward or data-gradient)."""

    __slots__ = ("buf", "dtype", "ks", "cout", "src_ch", "cout_tiles", "layout", "call")

    def __init__(self, buf, dtype, ks, cout, src_ch, cout_tiles, layout="std", call=None):
        self.buf, self.dtype, self.ks, self.cout, self.src_ch, self.cout_tiles = buf, dtype, ks, cout, list(src_ch), cout_tiles
        self.layout = layout  # 'std': vmg_conv_pack; 'ws': vmg_convws_pack (the weight-streaming 3x3 kernel, deep = 3)
        self.call = call      # (weight data_ptr, O, I, o0, on, src_off, src_ch, transpose_flip): what a plan entry needs to redo this pack


def pack_conv_weight(w: torch.Tensor, dtype: torch.dtype, src_ch: Optional[Sequence[int]] = None,
                     src_off: Optional[Sequence[int]] = None, o0: int = 0, on: Optional[int] = None,
                     transpose_flip: bool = False, out: Optional[torch.Tensor] = None, cout_tiles: Optional[int] = None, groups: int = 1) -> PackedConv:
    """w: fp32 (O, I, KS, KS) or (O, I).  Forward pack: K slices = src_off/src_ch over I, outputs O[o0:o0+on).
    Data-gradient pack (transpose_flip): outputs I[o0:o0+on), K = O[src_off[0]:+src_ch[0]).
    groups > 1: w is the weight of a GROUPED convolution (I = channels per group); the pack is its dense block-diagonal operator over
    groups * I input channels, so that one launch serves all groups."""
    hip.require_cuda(w)
    if w.dtype != torch.float32 or not w.is_contiguous():
        raise HipError("pack_conv_weight expects a contiguous fp32 weight")
    if w.dim() == 2:
        O, I, ks = w.shape[0], w.shape[1], 1
    elif w.dim() == 4 and w.shape[2] == w.shape[3]:
        O, I, ks = w.shape[0], w.shape[1], w.shape[2]
    else:
        raise HipError(f"unsupported weight shape {tuple(w.shape)}")
    kdim, odim = (O, I * groups) if transpose_flip else (I * groups, O)
    if src_ch is None:
        src_ch, src_off = [kdim], [0]
    if src_off is None:
        src_off, acc = [], 0
        for c in src_ch:
            src_off.append(acc)
            acc += c
    if on is None:
        on = odim - o0
    tiles = cout_tiles if cout_tiles else cout_tiles_for(on, dtype, ks)
    flag = (1 if transpose_flip else 0) | ((groups << 8) if groups > 1 else 0)
    code = hip.dtype_code(dtype)
    l = hip.lib()
    nbytes = l.vmg_conv_pack_bytes(code, ks, on, len(src_ch), _intarr(src_ch), tiles)
    if nbytes <= 0:
        raise HipError(f"vmg_conv_pack_bytes rejected channels {list(src_ch)} (must be multiples of 8)")
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    elif out.numel() * out.element_size() < nbytes:
        raise HipError("pack buffer too small")
    hip.check(l.vmg_conv_pack(code, w.data_ptr(), O, I, ks, o0, on, len(src_ch), _intarr(src_off), _intarr(src_ch),
                              flag, tiles, out.data_ptr(), hip.stream_ptr()), "vmg_conv_pack")
    return PackedConv(out, dtype, ks, on, src_ch, tiles, call=(w.data_ptr(), O, I, o0, on, list(src_off), list(src_ch), flag))


WS_128_BLOCKS = os.environ.get("VMG_WS_128", "1") != "0"  # A/B switch of the 128-channel-block weight-streaming route


def ws_eligible(cout: int, ks: int, dtype: torch.dtype, src_ch: Sequence[int]) -> int:
    """cout_tiles (9, 7 or 8) if the weight-streaming kernel covers this conv (bf16, 3x3, output channels in blocks of 144, 112 or 128,
    source channel counts that split into blocks of <= 160 channels that are multiples of 16), else 0."""
    if dtype != torch.bfloat16 or ks != 3:
        return 0
    for c in src_ch:
        parts = 1
        while c // parts > 160 or c % (8 * parts):
            parts += 1
            if parts > c // 8:
                return 0
        if (c // parts) % 16:
            return 0
    if cout % 144 == 0:
        return 9
    if cout % 112 == 0:
        return 7
    if cout % 128 == 0 and WS_128_BLOCKS:
        return 8  # (round 4: `upconv2`, 144 -> 256 before the second PixelShuffle: 128-channel blocks)
    return 0


def pack_conv_weight_ws(w: torch.Tensor, src_ch: Optional[Sequence[int]] = None, src_off: Optional[Sequence[int]] = None, o0: int = 0,
                        on: Optional[int] = None, transpose_flip: bool = False, cout_tiles: int = 9, out: Optional[torch.Tensor] = None,
                        groups: int = 1) -> PackedConv:
    """bf16 pack of a 3x3 weight (O, I, 3, 3) for the weight-streaming kernel (same slicing conventions as pack_conv_weight).
    out: an existing pack buffer of this very pack to rewrite in place."""
    hip.require_cuda(w)
    if w.dtype != torch.float32 or not w.is_contiguous() or w.dim() != 4 or w.shape[2] != 3 or w.shape[3] != 3:
        raise HipError("pack_conv_weight_ws expects a contiguous fp32 (O, I, 3, 3) weight")
    O, I = w.shape[0], w.shape[1]
    kdim, odim = (O, I * groups) if transpose_flip else (I * groups, O)
    if src_ch is None:
        src_ch, src_off = [kdim], [0]
    if src_off is None:
        src_off, acc = [], 0
        for c in src_ch:
            src_off.append(acc)
            acc += c
    if on is None:
        on = odim - o0
    flag = (1 if transpose_flip else 0) | ((groups << 8) if groups > 1 else 0)
    l = hip.lib()
    nbytes = l.vmg_convws_pack_bytes(on, len(src_ch), _intarr(src_ch), cout_tiles)
    if nbytes <= 0:
        raise HipError(f"vmg_convws_pack_bytes rejected channels {list(src_ch)} / cout_tiles {cout_tiles}")
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    elif out.numel() * out.element_size() < nbytes:
        raise HipError("pack buffer too small")
    hip.check(l.vmg_convws_pack(w.data_ptr(), O, I, o0, on, len(src_ch), _intarr(src_off), _intarr(src_ch), flag,
                                cout_tiles, out.data_ptr(), hip.stream_ptr()), "vmg_convws_pack")
    return PackedConv(out, torch.bfloat16, 3, on, src_ch, cout_tiles, layout="ws",
                      call=(w.data_ptr(), O, I, o0, on, list(src_off), list(src_ch), flag))


# vmg_conv_desc as one struct format (checked against the ctypes layout at import): field order of hip.ConvDesc
_CONV_FMT = struct.Struct("=8i4Q4q4iQQQqQQqQqiffiiii4x")
assert _CONV_FMT.size == ctypes.sizeof(ConvDesc), "vmg_conv_desc layout drifted from kernels._CONV_FMT"
_CONV_DESC = ConvDesc()  # reused: vmg_conv_fwd copies what it needs before it returns (single-threaded host side)
_CONV_PACK = _CONV_FMT.pack_into


def _pix_stride(t: torch.Tensor) -> int:
    """Pixel stride (elements) of a channels-last tensor whose leading dims are dense over pixels."""
    st = t.stride()
    if st[-1] != 1:
        raise HipError("channels must be the fastest dimension")
    if t.is_contiguous():  # the common case: one call instead of a size/stride pair per dimension
        return t.shape[-1]
    sh = t.shape
    ps = st[-2]
    # leading dims must be a dense pixel enumeration with that stride
    expect = ps
    for d in range(len(sh) - 2, -1, -1):
        if sh[d] != 1 and st[d] != expect:
            raise HipError(f"tensor of shape {tuple(sh)} / strides {st} is not a dense pixel array")
        expect *= sh[d]
    return ps


def conv_forward(srcs: Sequence[torch.Tensor], pw: PackedConv, bias: Optional[torch.Tensor], N: int, H: int, W: int,
                 act: int = hip.ACT_NONE, slope: float = 0.0, alpha: float = 1.0, res: Optional[torch.Tensor] = None,
                 aux: Optional[torch.Tensor] = None, actgrad: int = 0, pixel_shuffle: bool = False,
                 out: Optional[torch.Tensor] = None, out_pre: Optional[torch.Tensor] = None, want_pre: bool = False,
                 mt: int = 0, deep: int = 0) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Runs vmg_conv_fwd.  srcs: channels-last tensors (..., C_s) covering N*H*W pixels each (channel slices of
    wider tensors are fine).  Returns (out, out_pre)."""
    if len(srcs) != len(pw.src_ch):
        raise HipError(f"conv expects {len(pw.src_ch)} sources, got {len(srcs)}")
    x0 = srcs[0]
    hip.require_cuda(*srcs, bias, res, aux, out)
    dt = x0.dtype
    if dt != pw.dtype:
        raise HipError(f"activation dtype {dt} != packed weight dtype {pw.dtype}")
    M = N * H * W
    nsrc = len(srcs)
    if (pw.layout == "ws") != (deep == 3):
        deep = 3 if pw.layout == "ws" else 0  # the packed layout decides the kernel
    sp, sps, sch = [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]
    for i, s in enumerate(srcs):
        if s.dtype != dt or s.shape[-1] != pw.src_ch[i] or s.numel() // s.shape[-1] != M:
            raise HipError(f"source {i}: shape {tuple(s.shape)} / dtype {s.dtype} does not match conv (M={M}, C={pw.src_ch[i]}, {dt})")
        sp[i], sps[i], sch[i] = s.data_ptr(), _pix_stride(s), pw.src_ch[i]
    bias_p = 0
    if bias is not None:
        if bias.dtype != torch.float32 or bias.numel() != pw.cout or not bias.is_contiguous():
            raise HipError("bias must be contiguous fp32 of length Cout")
        bias_p = bias.data_ptr()
    if pixel_shuffle:
        oshape = (N, 2 * H, 2 * W, pw.cout // 4)
    else:
        oshape = (N, H, W, pw.cout)
    if out is None:
        out = torch.empty(oshape, dtype=dt, device=x0.device)
    elif out.dtype != dt or out.numel() // out.shape[-1] != (4 * M if pixel_shuffle else M) or out.shape[-1] != oshape[-1]:
        raise HipError(f"bad output tensor {tuple(out.shape)} for conv output {oshape}")
    out_ps = _pix_stride(out)
    if want_pre and out_pre is None:
        if pixel_shuffle:
            raise HipError("out_pre is not available with pixel_shuffle")
        out_pre = torch.empty(oshape, dtype=dt, device=x0.device)
    pre_p = 0
    if out_pre is not None:
        if _pix_stride(out_pre) != out_ps or out_pre.shape[-1] != pw.cout or out_pre.dtype != dt:
            raise HipError("out_pre must have the layout of out")
        pre_p = out_pre.data_ptr()
    extra = [0, 0, 0, 0]  # res, res_ps, aux, aux_ps
    for k, (name, t) in enumerate((("res", res), ("aux", aux))):
        if t is None:
            continue
        if pixel_shuffle:
            raise HipError(f"{name} is not supported together with pixel_shuffle")
        if t.dtype != dt or t.shape[-1] != pw.cout or t.numel() // t.shape[-1] != M:
            raise HipError(f"{name}: shape {tuple(t.shape)} does not match conv output")
        extra[2 * k], extra[2 * k + 1] = t.data_ptr(), _pix_stride(t)
    # the descriptor is filled with ONE struct.pack_into (30 ctypes attribute stores cost ~9 us per conv, 5 ms per train step)
    d = _CONV_DESC
    _CONV_PACK(d, 0, hip.dtype_code(dt), pw.ks, pw.cout_tiles, N, H, W, pw.cout, nsrc, sp[0], sp[1], sp[2], sp[3], sps[0], sps[1], sps[2], sps[3],
               sch[0], sch[1], sch[2], sch[3], pw.buf.data_ptr(), bias_p, out.data_ptr(), out_ps, pre_p, extra[0], extra[1], extra[2], extra[3],
               act, slope, alpha, actgrad, int(pixel_shuffle), mt, deep)
    hip.check(hip.lib().vmg_conv_fwd(ctypes.byref(d), hip.stream_ptr()), "vmg_conv_fwd")
    return out, out_pre


def _parr(ts):
    return (ctypes.c_void_p * max(1, len(ts)))(*[(t.data_ptr() if t is not None else None) for t in ts])


CHAIN_STATS = {"fwd": {}, "bwd": {}}  # chain calls by the `deep` route their block convolutions took (3: weight-streaming, 2: K-split, ...): read by tests


def resblock_chain_forward(srcs: Sequence[torch.Tensor], pw0: PackedConv, b0: torch.Tensor, slope0: float, deep0: int, pw1: Sequence[PackedConv],
                           b1: Sequence[torch.Tensor], pw2: Sequence[PackedConv], b2: Sequence[torch.Tensor], r_scaling: float, deep: int,
                           own_output: bool = False):
    """ResidualBlocksWithInputConv forward as ONE C call (vmg_resblock_chain_fwd).  Returns (ys, ts): nblk + 1 block outputs and nblk
    ReLU outputs, all (N, H, W, C)."""
    x0 = srcs[0]
    hip.require_cuda(*srcs, b0, *b1, *b2)
    N, H, W = x0.shape[0], x0.shape[1], x0.shape[2]
    M = N * H * W
    C, nblk, dt = pw0.cout, len(pw1), x0.dtype
    if len(srcs) != len(pw0.src_ch) or any(s.dtype != dt or s.shape[-1] != c or s.numel() // s.shape[-1] != M for s, c in zip(srcs, pw0.src_ch)):
        raise HipError("resblock_chain: sources do not match conv0's pack")
    if any(p.cout != C or p.src_ch != [C] or p.dtype != dt or p.layout != pw1[0].layout or p.cout_tiles != pw1[0].cout_tiles for p in list(pw1) + list(pw2)) or \
            pw0.dtype != dt or len(pw2) != nblk or len(b1) != nblk or len(b2) != nblk:
        raise HipError("resblock_chain: block packs must be C -> C packs of one layout")
    for b in [b0] + list(b1) + list(b2):
        if b.dtype != torch.float32 or b.numel() != C or not b.is_contiguous():
            raise HipError("resblock_chain: biases must be contiguous fp32 of length C")
    blk = torch.empty((2 * nblk + 1, N, H, W, C), dtype=dt, device=x0.device).unbind(0)  # ONE allocation: 31 torch.empty calls cost 0.1 ms of host time per chain
    ys, ts = list(blk[:nblk + 1]), list(blk[nblk + 1:])
    if own_output:  # the chain's output in its own allocation: the caller drops the intermediates (activation recompute)
        ys[-1] = torch.empty((N, H, W, C), dtype=dt, device=x0.device)
    d = hip.ChainDesc()
    d.dtype, d.N, d.H, d.W, d.C, d.nblk, d.nsrc = hip.dtype_code(dt), N, H, W, C, nblk, len(srcs)
    for i, s in enumerate(srcs):
        d.src[i], d.src_ps[i], d.src_ch[i] = s.data_ptr(), _pix_stride(s), pw0.src_ch[i]
    d.packed0, d.bias0, d.slope0 = pw0.buf.data_ptr(), b0.data_ptr(), slope0
    d.cout_tiles0, d.deep0 = pw0.cout_tiles, 3 if pw0.layout == "ws" else deep0
    keep = [_parr([p.buf for p in pw1]), _parr(list(b1)), _parr([p.buf for p in pw2]), _parr(list(b2)), _parr(ys), _parr(ts)]
    d.packed1, d.bias1, d.packed2, d.bias2, d.y, d.t = keep
    d.r_scaling = r_scaling
    d.cout_tiles = pw1[0].cout_tiles if nblk else pw0.cout_tiles
    d.deep = (3 if pw1[0].layout == "ws" else deep) if nblk else 0
    hip.check(hip.lib().vmg_resblock_chain_fwd(ctypes.byref(d), hip.stream_ptr()), "vmg_resblock_chain_fwd")
    CHAIN_STATS["fwd"][d.deep] = CHAIN_STATS["fwd"].get(d.deep, 0) + 1
    return ys, ts


def resblock_chain_backward(g: torch.Tensor, ts: Sequence[torch.Tensor], pd1: Sequence[PackedConv], pd2: Sequence[PackedConv], r_scaling: float,
                            deep: int):
    """Data-gradient sweep of the residual blocks (vmg_resblock_chain_bwd).  g: gradient of the chain output; pd1 / pd2: data-gradient
    packs of conv1 / conv2 per block.  Returns (g_ys, g_ts): g_ys[k] = gradient of y_k (g_ys[nblk] is g), g_ts[k] of conv1_k's pre-activation."""
    hip.require_cuda(g, *ts)
    nblk = len(ts)
    N, H, W, C = g.shape
    g = g.contiguous()
    if nblk == 0:
        return [g], []
    blk = torch.empty((2 * nblk, N, H, W, C), dtype=g.dtype, device=g.device).unbind(0)
    gys, gts = list(blk[:nblk]) + [g], list(blk[nblk:])
    d = hip.ChainDesc()
    d.dtype, d.N, d.H, d.W, d.C, d.nblk, d.nsrc = hip.dtype_code(g.dtype), N, H, W, C, nblk, 1
    keep = [_parr([p.buf for p in pd1]), _parr([p.buf for p in pd2]), _parr(list(ts)), _parr(gys), _parr(gts)]
    d.packed1, d.packed2, d.t, d.g_y, d.g_t = keep
    d.r_scaling = r_scaling
    d.cout_tiles = pd1[0].cout_tiles
    d.deep = 3 if pd1[0].layout == "ws" else deep
    hip.check(hip.lib().vmg_resblock_chain_bwd(ctypes.byref(d), hip.stream_ptr()), "vmg_resblock_chain_bwd")
    CHAIN_STATS["bwd"][d.deep] = CHAIN_STATS["bwd"].get(d.deep, 0) + 1
    return gys, gts


def conv_wgrad(x: torch.Tensor, dy: torch.Tensor, dW: torch.Tensor, db: Optional[torch.Tensor], ks: int, N: int, H: int,
               W: int, scale: float = 1.0, o0: int = 0, i0: int = 0):
    """dW (fp32, (O_total, I_total, ks, ks) or (O_total, I_total)) += scale * wgrad(x, dy); db += scale * sum(dy).
    x (..., Cin) and dy (..., Cout) are channels-last tensors over the same N*H*W pixels (channel slices allowed)."""
    hip.require_cuda(x, dy, dW, db)
    if dW.dtype != torch.float32 or not dW.is_contiguous() or (db is not None and (db.dtype != torch.float32 or not db.is_contiguous())):
        raise HipError("parameter gradients must be contiguous fp32")
    if x.dtype != dy.dtype:
        raise HipError("x and dy must share a dtype")
    M = N * H * W
    Cin, Cout = x.shape[-1], dy.shape[-1]
    if x.numel() // Cin != M or dy.numel() // Cout != M:
        raise HipError("x / dy do not cover N*H*W pixels")
    O_total, I_total = dW.shape[0], dW.shape[1]
    kk = 1 if dW.dim() == 2 else dW.shape[2]
    if kk != ks or o0 + Cout > O_total or i0 + Cin > I_total or (db is not None and db.numel() != O_total):
        raise HipError(f"gradient tensor {tuple(dW.shape)} does not match conv (ks={ks}, Cout={Cout}+{o0}, Cin={Cin}+{i0})")
    hip.check(hip.lib().vmg_conv_wgrad(hip.dtype_code(x.dtype), ks, N, H, W, x.data_ptr(), _pix_stride(x), Cin, dy.data_ptr(),
                                       _pix_stride(dy), Cout, dW.data_ptr(), I_total, o0, i0,
                                       db.data_ptr() if db is not None else None, scale, hip.stream_ptr()), "vmg_conv_wgrad")


def act_backward(dy: torch.Tensor, ref: torch.Tensor, act: int, slope: float, alpha: float) -> torch.Tensor:
    """dy * alpha * act'(ref); ref = activation output (RELU/LRELU) or pre-activation (GELU)."""
    hip.require_cuda(dy, ref)
    if dy.shape != ref.shape or dy.dtype != ref.dtype or not dy.is_contiguous() or not ref.is_contiguous():
        raise HipError("act_backward: dy / ref must be contiguous tensors of one shape and dtype")
    out = torch.empty_like(dy)
    hip.check(hip.lib().vmg_act_bwd(hip.dtype_code(dy.dtype), dy.data_ptr(), ref.data_ptr(), out.data_ptr(), dy.numel(), act, slope,
                                    alpha, hip.stream_ptr()), "vmg_act_bwd")
    return out


def pixel_shuffle(x: torch.Tensor, N: int, H: int, W: int) -> torch.Tensor:
    """(N,H,W,4c) -> (N,2H,2W,c), torch PixelShuffle(2) order."""
    hip.require_cuda(x)
    c4 = x.shape[-1]
    if c4 % 4 or x.numel() != N * H * W * c4 or not x.is_contiguous():
        raise HipError("pixel_shuffle: bad input")
    out = torch.empty((N, 2 * H, 2 * W, c4 // 4), dtype=x.dtype, device=x.device)
    hip.check(hip.lib().vmg_pixel_shuffle(hip.dtype_code(x.dtype), x.data_ptr(), out.data_ptr(), N, H, W, c4 // 4, 0, hip.stream_ptr()),
              "vmg_pixel_shuffle")
    return out


def frame_gather(src: torch.Tensor, idx: torch.Tensor, n_dst: int, frame_shape) -> torch.Tensor:
    """dst frame f = sum_k src frame idx[f, k] (vmg_frame_gather): src contiguous, its leading dims flattened into frames of prod(frame_shape)
    elements; idx (n_dst, 1 or 2) int32 on the device, every entry < number of source frames (the CALLER's contract: the table is device data),
    < 0 = no term.  Returns (n_dst, *frame_shape)."""
    hip.require_cuda(src, idx)
    fe = 1
    for d in frame_shape:
        fe *= int(d)
    if not src.is_contiguous() or src.numel() % fe or idx.dtype != torch.int32 or idx.dim() != 2 or idx.shape[0] != n_dst or idx.shape[1] not in (1, 2) or not idx.is_contiguous():
        raise HipError("frame_gather: contiguous src of whole frames and a contiguous int32 (n_dst, 1|2) index table expected")
    dst = torch.empty((n_dst, *frame_shape), dtype=src.dtype, device=src.device)
    hip.check(hip.lib().vmg_frame_gather(hip.dtype_code(src.dtype), src.data_ptr(), dst.data_ptr(), idx.data_ptr(), fe, src.numel() // fe, n_dst, idx.shape[1],
                                         hip.stream_ptr()), "vmg_frame_gather")
    return dst


def sum_n(ts: Sequence[torch.Tensor]) -> torch.Tensor:
    """Sum of 2..4 contiguous tensors of one shape and dtype (bf16 / fp32), fp32 sum, one rounding (vmg_sum_n)."""
    hip.require_cuda(*ts)
    t0 = ts[0]
    vn = 8 if t0.dtype == torch.bfloat16 else 4
    if not 2 <= len(ts) <= 4 or t0.numel() % vn or any(t.shape != t0.shape or t.dtype != t0.dtype or not t.is_contiguous() for t in ts):
        raise HipError("sum_n: 2..4 contiguous tensors of one shape and dtype, whole 16-byte vectors")
    out = torch.empty_like(t0)
    ptrs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    hip.check(hip.lib().vmg_sum_n(hip.dtype_code(t0.dtype), ptrs, len(ts), out.data_ptr(), t0.numel(), hip.stream_ptr()), "vmg_sum_n")
    return out


class _AccPool:
    """fp32 scatter accumulators at rest are ZERO (cast_clear leaves them so): a buffer is taken for one scatter, rounded + cleared, given back -- no fill pass per
    use.  A buffer that is not given back (an exception in between) is simply dropped.  One stream: every user runs on the current stream, in order."""

    def __init__(self):
        self.free = {}

    def take(self, shape, device) -> torch.Tensor:
        lst = self.free.get((tuple(shape), str(device)))
        return lst.pop() if lst else torch.zeros(tuple(shape), dtype=torch.float32, device=device)

    def give(self, buf: torch.Tensor) -> None:
        # (never dropped on its own: a captured hipGraph has the addresses of the buffers it used baked into its kernel nodes -- the same reason
        #  PackPlan retires its tables instead of freeing them; clear() is the caller's explicit decision)
        self.free.setdefault((tuple(buf.shape), str(buf.device)), []).append(buf)

    def clear(self) -> None:
        self.free.clear()


ACC_POOL = _AccPool()


def cast_clear(acc: torch.Tensor, dtype: torch.dtype, add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(acc [+ add]) rounded to `dtype` once; acc (contiguous fp32) is ZERO afterwards (vmg_cast_clear)."""
    hip.require_cuda(acc, add)
    if acc.dtype != torch.float32 or not acc.is_contiguous() or acc.numel() % 4 or (add is not None and (add.dtype != dtype or add.shape != acc.shape or not add.is_contiguous())):
        raise HipError("cast_clear: contiguous fp32 accumulator (a multiple of 4 elements); `add` of the output dtype and the same shape")
    out = torch.empty(acc.shape, dtype=dtype, device=acc.device)
    hip.check(hip.lib().vmg_cast_clear(hip.dtype_code(dtype), acc.data_ptr(), add.data_ptr() if add is not None else None, out.data_ptr(), acc.numel(), hip.stream_ptr()),
              "vmg_cast_clear")
    return out


def pair_steps(mode: int, steps: Sequence[torch.Tensor], a: torch.Tensor, b: Optional[torch.Tensor], n: int, t: int) -> None:
    """vmg_pair_steps: the t step tensors of the lock-step recurrence ((2n, *frame) each, separate allocations) <-> a, b = the two sweeps' features
    (n, t, *frame) in frame order.  mode 0: steps -> a, b; mode 1: a, b -> steps; mode 2: a[i, f] = steps[t-1-f][i] + steps[f][n+i] (b unused)."""
    hip.require_cuda(a, b, *steps)
    if len(steps) != t or t < 1 or a.shape[0] != n or a.shape[1] != t or not a.is_contiguous():
        raise HipError("pair_steps: t step tensors and a contiguous (n, t, ...) tensor expected")
    fe = a.numel() // (n * t)
    for st in steps:
        if st.dtype != a.dtype or st.numel() != 2 * n * fe or not st.is_contiguous():
            raise HipError("pair_steps: every step tensor must be contiguous (2n, *frame) of the same dtype")
    if mode != 2 and (b is None or b.shape != a.shape or b.dtype != a.dtype or not b.is_contiguous()):
        raise HipError("pair_steps: the second (n, t, ...) tensor must match the first")
    ptrs = (ctypes.c_void_p * t)(*[st.data_ptr() for st in steps])
    hip.check(hip.lib().vmg_pair_steps(hip.dtype_code(a.dtype), mode, ptrs, a.data_ptr(), b.data_ptr() if b is not None else None, n, t, fe, hip.stream_ptr()),
              "vmg_pair_steps")


def pixel_unshuffle_actgrad(dy: torch.Tensor, ref: Optional[torch.Tensor], N: int, H: int, W: int, act: int, slope: float, alpha: float) -> torch.Tensor:
    """(N,2H,2W,c) gradient (and activation reference) -> (N,H,W,4c) pre-activation gradient of a PixelShuffle conv, one pass."""
    hip.require_cuda(dy, ref)
    c = dy.shape[-1]
    if dy.numel() != 4 * N * H * W * c or not dy.is_contiguous() or (ref is not None and (ref.shape != dy.shape or ref.dtype != dy.dtype or not ref.is_contiguous())):
        raise HipError("pixel_unshuffle_actgrad: bad input")
    out = torch.empty((N, H, W, 4 * c), dtype=dy.dtype, device=dy.device)
    hip.check(hip.lib().vmg_pixel_unshuffle_actgrad(hip.dtype_code(dy.dtype), dy.data_ptr(), ref.data_ptr() if ref is not None else None, out.data_ptr(),
                                                    N, H, W, c, act if ref is not None else hip.ACT_NONE, slope, alpha, hip.stream_ptr()),
              "vmg_pixel_unshuffle_actgrad")
    return out


def pixel_unshuffle(x: torch.Tensor, N: int, H: int, W: int) -> torch.Tensor:
    """(N,2H,2W,c) -> (N,H,W,4c): inverse of pixel_shuffle."""
    hip.require_cuda(x)
    c = x.shape[-1]
    if x.numel() != N * 4 * H * W * c or not x.is_contiguous():
        raise HipError("pixel_unshuffle: bad input")
    out = torch.empty((N, H, W, 4 * c), dtype=x.dtype, device=x.device)
    hip.check(hip.lib().vmg_pixel_shuffle(hip.dtype_code(x.dtype), x.data_ptr(), out.data_ptr(), N, H, W, c, 1, hip.stream_ptr()),
              "vmg_pixel_shuffle")
    return out


def layernorm_forward(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5, want_stats: bool = True):
    hip.require_cuda(x, w, b)
    C = x.shape[-1]
    if not x.is_contiguous() or w.dtype != torch.float32 or b.dtype != torch.float32 or w.numel() != C or b.numel() != C:
        raise HipError("layernorm: x must be contiguous, w/b fp32 of length C")
    M = x.numel() // C
    y = torch.empty_like(x)
    mean = torch.empty(M, dtype=torch.float32, device=x.device) if want_stats else None
    rstd = torch.empty(M, dtype=torch.float32, device=x.device) if want_stats else None
    hip.check(hip.lib().vmg_layernorm_fwd(hip.dtype_code(x.dtype), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(),
                                          mean.data_ptr() if want_stats else None, rstd.data_ptr() if want_stats else None,
                                          M, C, eps, hip.stream_ptr()), "vmg_layernorm_fwd")
    return y, mean, rstd


def _grad_pair(C: int, device, into):
    """(dw, db) accumulators of a LayerNorm backward: fresh zeros, or the caller's fp32 buffers (param.grad) to add into."""
    if into is None:
        return torch.zeros(C, dtype=torch.float32, device=device), torch.zeros(C, dtype=torch.float32, device=device)
    dw, db = into
    for t in (dw, db):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != C or not t.is_cuda:
            raise HipError("LayerNorm gradient buffers must be contiguous fp32 (C,) on the GPU")
    return dw, db


def layernorm_backward(dy, x: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor, w: torch.Tensor, into=None, add=None):
    """into = (dw, db): the kernel's atomic sums are ADDED to these buffers (param.grad) instead of to fresh zero tensors.
    add: a gradient of x's shape that is summed into dx by the kernel (the skip-connection gradient).
    dy: one gradient, or a list of up to five gradients of the LayerNorm output that the kernel sums on the way in."""
    C = x.shape[-1]
    M = x.numel() // C
    if isinstance(dy, (list, tuple)):
        dys = [d.contiguous() for d in dy]
        hip.require_cuda(x, mean, rstd, w, add, *dys)
        if not 1 <= len(dys) <= 5 or any(d.shape != x.shape or d.dtype != x.dtype for d in dys):
            raise HipError("layernorm_backward: 1..5 output gradients of x's shape and dtype expected")
        dx = torch.empty_like(x)
        dw, db = _grad_pair(C, x.device, into)
        if add is not None:
            if add.shape != x.shape or add.dtype != x.dtype:
                raise HipError("layernorm_backward: add must have x's shape and dtype")
            add = add.contiguous()
        hip.check(hip.lib().vmg_layernorm_bwd_multi(hip.dtype_code(x.dtype), len(dys), _ptrs(dys), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), w.data_ptr(),
                                                    add.data_ptr() if add is not None else None, dx.data_ptr(), dw.data_ptr(), db.data_ptr(), M, C,
                                                    hip.stream_ptr()), "vmg_layernorm_bwd_multi")
        return dx, dw, db
    hip.require_cuda(dy, x, mean, rstd, w, add)
    dy = dy.contiguous()
    dx = torch.empty_like(x)
    dw, db = _grad_pair(C, x.device, into)
    if add is not None:
        if add.shape != x.shape or add.dtype != x.dtype:
            raise HipError("layernorm_backward: add must have x's shape and dtype")
        add = add.contiguous()
        hip.check(hip.lib().vmg_layernorm_bwd_add(hip.dtype_code(x.dtype), dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), w.data_ptr(),
                                                  add.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), M, C, hip.stream_ptr()),
                  "vmg_layernorm_bwd_add")
        return dx, dw, db
    hip.check(hip.lib().vmg_layernorm_bwd(hip.dtype_code(x.dtype), dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                          w.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), M, C, hip.stream_ptr()),
              "vmg_layernorm_bwd")
    return dx, dw, db


MORPH_FUSED_CP = (144, 112, 64, 32, 16)


def morph_fused_ok(x: torch.Tensor, chunk: int, Cp: int) -> bool:
    """Shapes the fused MorphFC kernel (vmg_morphfc_fwd) covers: bf16, chunk 8 or 16, Cp instantiated, C a multiple of 8."""
    return x.dtype == torch.bfloat16 and chunk in (8, 16) and Cp in MORPH_FUSED_CP and x.shape[-1] % 8 == 0 and Cp % chunk == 0


def morphfc_forward(x: torch.Tensor, axis: str, chunk: int, Cp: int, pw: PackedConv, bias: Optional[torch.Tensor], relu: bool, in_scale: float,
                    out_scale: float, mask: Optional[torch.Tensor] = None, want_tokens: bool = False):
    """x (B,T,H,W,C) contiguous bf16 -> the branch output in the same layout (vmg_morphfc_fwd); mask: the data-gradient form.  want_tokens:
    also returns the token matrix (rows, Cp) the GEMM multiplied (the weight gradient's operand), else None."""
    hip.require_cuda(x, bias, mask)
    B, T, H, W, C = x.shape
    if not x.is_contiguous() or (mask is not None and (not mask.is_contiguous() or mask.shape != x.shape or mask.dtype != x.dtype)):
        raise HipError("morphfc: contiguous x (and mask of the same shape) expected")
    if pw.layout != "std" or pw.ks != 1 or pw.cout != Cp or pw.src_ch != [Cp] or pw.cout_tiles != (Cp + 15) // 16 or pw.dtype != torch.bfloat16:
        raise HipError("morphfc: the pack must be a (Cp, Cp) 1x1 pack with all output tiles in one block")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != Cp or not bias.is_contiguous()):
        raise HipError("morphfc: bias must be contiguous fp32 of length Cp")
    out = torch.empty_like(x)
    ax = 0 if axis == "h" else 1
    tok = None
    if want_tokens:
        rows = hip.lib().vmg_morphfc_token_rows(ax, chunk, B * T, H, W)
        if rows <= 0:
            raise HipError("morphfc: bad token geometry")
        tok = torch.empty((rows, Cp), dtype=x.dtype, device=x.device)
    hip.check(hip.lib().vmg_morphfc_fwd(ax, chunk, x.data_ptr(), mask.data_ptr() if mask is not None else None, pw.buf.data_ptr(),
                                        bias.data_ptr() if bias is not None else None, out.data_ptr(), tok.data_ptr() if tok is not None else None,
                                        B * T, H, W, C, Cp, pw.cout_tiles, 1 if relu else 0, in_scale, out_scale, hip.stream_ptr()), "vmg_morphfc_fwd")
    return out, tok


def morph_tokens_gather(x: torch.Tensor, axis: str, chunk: int, Cp: int, ld: Optional[int] = None) -> torch.Tensor:
    """x (B,T,H,W,C) contiguous -> the MorphFC token matrix (rows, ld) of the reference's pad + rearrange chain (vmg_morph_tokens_gather): row
    (group, k), feature p*S + s = x[position p of the group, channel k*S + s], zeros for padding positions / channels and for features >= Cp."""
    hip.require_cuda(x)
    B, T, H, W, C = x.shape
    if not x.is_contiguous():
        raise HipError("morph_tokens_gather: contiguous (B,T,H,W,C) expected")
    ld = Cp if ld is None else ld
    ax = 0 if axis == "h" else 1
    rows = hip.lib().vmg_morph_token_rows(ax, chunk, B * T, H, W)
    tok = torch.empty((rows, ld), dtype=x.dtype, device=x.device)
    hip.check(hip.lib().vmg_morph_tokens_gather(hip.dtype_code(x.dtype), ax, chunk, x.data_ptr(), tok.data_ptr(), B * T, H, W, C, Cp, ld, hip.stream_ptr()),
              "vmg_morph_tokens_gather")
    return tok


def morph_tokens_scatter(tok: torch.Tensor, axis: str, chunk: int, Cp: int, shape) -> torch.Tensor:
    """The inverse layout + crop: tok (rows, ld >= Cp), row stride ld -> (B,T,H,W,C) (vmg_morph_tokens_scatter)."""
    hip.require_cuda(tok)
    B, T, H, W, C = shape
    ax = 0 if axis == "h" else 1
    rows = hip.lib().vmg_morph_token_rows(ax, chunk, B * T, H, W)
    if tok.dim() != 2 or tok.shape[0] != rows or tok.shape[1] < Cp or tok.stride(1) != 1:
        raise HipError(f"morph_tokens_scatter: token matrix ({rows}, >= {Cp}) expected, got {tuple(tok.shape)}")
    out = torch.empty((B, T, H, W, C), dtype=tok.dtype, device=tok.device)
    hip.check(hip.lib().vmg_morph_tokens_scatter(hip.dtype_code(tok.dtype), ax, chunk, tok.data_ptr(), out.data_ptr(), B * T, H, W, C, Cp, tok.stride(0),
                                                 hip.stream_ptr()), "vmg_morph_tokens_scatter")
    return out


def _win3d_geom(q, kv, table, heads, wt):
    B, D, H, W, C = q.shape
    if tuple(kv.shape) != (B, D, H, W, 2 * C) or kv.dtype != q.dtype or not q.is_contiguous() or not kv.is_contiguous():
        raise HipError("win3d_attn: contiguous q (B,D,H,W,C) and kv (B,D,H,W,2C) of one dtype expected")
    if table.dtype != torch.float32 or not table.is_contiguous() or tuple(table.shape) != ((2 * wt - 1) * 225, heads):
        raise HipError(f"win3d_attn: the bias table must be contiguous fp32 ((2*{wt}-1)*225, {heads})")
    nwin = B * ((D + wt - 1) // wt) * ((H + 7) // 8) * ((W + 7) // 8)
    return B, D, H, W, C, nwin


def win3d_attn_forward(q, kv, bq, bkv, table, heads: int, wt: int, shift):
    hip.require_cuda(q, kv, bq, bkv, table)
    B, D, H, W, C, nwin = _win3d_geom(q, kv, table, heads, wt)
    out = torch.empty_like(q)
    lse = torch.empty((nwin, heads, wt * 64), dtype=torch.float32, device=q.device)
    pz = lambda t: t.data_ptr() if t is not None else None
    hip.check(hip.lib().vmg_win3d_attn_fwd(hip.dtype_code(q.dtype), q.data_ptr(), kv.data_ptr(), pz(bq), pz(bkv), table.data_ptr(), out.data_ptr(), lse.data_ptr(),
                                           B, D, H, W, C, heads, wt, shift[0], shift[1], shift[2], hip.stream_ptr()), "vmg_win3d_attn_fwd")
    return out, lse


def win3d_attn_backward(q, kv, bq, bkv, table, out, lse, dout, heads: int, wt: int, shift, into=None):
    """into = (dbq, dbkv): existing contiguous fp32 buffers (the biases' .grad) the kernel ADDS its bias gradients to, instead of fresh zeros."""
    B, D, H, W, C, nwin = _win3d_geom(q, kv, table, heads, wt)
    dout = dout.contiguous()
    dq, dkv = torch.empty_like(q), torch.empty_like(kv)
    dtable = torch.zeros_like(table)
    if into is not None:
        dbq, dbkv = into
        hip.require_cuda(dbq, dbkv)
        if dbq.dtype != torch.float32 or dbkv.dtype != torch.float32 or dbq.numel() != C or dbkv.numel() != 2 * C or not dbq.is_contiguous() or not dbkv.is_contiguous():
            raise HipError("win3d_attn_backward: bias gradient buffers must be contiguous fp32 of C and 2C elements")
    else:
        dbq = torch.zeros(C, dtype=torch.float32, device=q.device) if bq is not None else None
        dbkv = torch.zeros(2 * C, dtype=torch.float32, device=q.device) if bkv is not None else None
    pz = lambda t: t.data_ptr() if t is not None else None
    # per-workgroup table gradients + an ordered reduce launch instead of float atomics from every window onto the same table (kernel doc)
    ws = torch.empty(int(hip.lib().vmg_win3d_attn_bwd_ws_bytes(B, D, H, W, heads, wt)), dtype=torch.uint8, device=q.device)
    hip.check(hip.lib().vmg_win3d_attn_bwd(hip.dtype_code(q.dtype), q.data_ptr(), kv.data_ptr(), pz(bq), pz(bkv), table.data_ptr(), out.data_ptr(), lse.data_ptr(),
                                           dout.data_ptr(), dq.data_ptr(), dkv.data_ptr(), dtable.data_ptr(), pz(dbq), pz(dbkv), ws.data_ptr(), B, D, H, W, C, heads, wt,
                                           shift[0], shift[1], shift[2], hip.stream_ptr()), "vmg_win3d_attn_bwd")
    return dq, dkv, dtable, dbq, dbkv


def maxpool_forward(x: torch.Tensor, f: int):
    """Non-overlapping f x f max pooling of a contiguous channels-last (n,h,w,c) tensor -> (y, idx)."""
    hip.require_cuda(x)
    n, h, w, c = x.shape
    if not x.is_contiguous() or h % f or w % f:
        raise HipError("maxpool: contiguous (n,h,w,c) with h, w multiples of the window expected")
    y = torch.empty((n, h // f, w // f, c), dtype=x.dtype, device=x.device)
    idx = torch.empty((n, h // f, w // f, c), dtype=torch.uint8, device=x.device)
    hip.check(hip.lib().vmg_maxpool_fwd(hip.dtype_code(x.dtype), x.data_ptr(), y.data_ptr(), idx.data_ptr(), n, h, w, c, f, hip.stream_ptr()), "vmg_maxpool_fwd")
    return y, idx


def maxpool_backward(dy: torch.Tensor, idx: torch.Tensor, f: int) -> torch.Tensor:
    hip.require_cuda(dy, idx)
    n, ho, wo, c = dy.shape
    dy = dy.contiguous()
    dx = torch.empty((n, ho * f, wo * f, c), dtype=dy.dtype, device=dy.device)
    hip.check(hip.lib().vmg_maxpool_bwd(hip.dtype_code(dy.dtype), dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), n, ho * f, wo * f, c, f, hip.stream_ptr()),
              "vmg_maxpool_bwd")
    return dx


def avgpool2(x: torch.Tensor) -> torch.Tensor:
    """F.avg_pool2d(x, 2, 2) on a contiguous channels-last (n, h, w, c) tensor."""
    hip.require_cuda(x)
    n, h, w, c = x.shape
    if not x.is_contiguous() or h < 2 or w < 2:
        raise HipError("avgpool2: contiguous (n,h,w,c) with h, w >= 2 expected")
    y = torch.empty((n, h // 2, w // 2, c), dtype=x.dtype, device=x.device)
    hip.check(hip.lib().vmg_avgpool2_nhwc(hip.dtype_code(x.dtype), x.data_ptr(), y.data_ptr(), n, h, w, c, hip.stream_ptr()), "vmg_avgpool2_nhwc")
    return y


def spy_prep(img: torch.Tensor, mean: torch.Tensor, std: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """img (n,3,h,w) fp32 contiguous -> (n,h,w,8) `dtype`: (img - mean) / std in channels 0..2, zeros behind (vmg_spy_prep)."""
    hip.require_cuda(img, mean, std)
    if img.dtype != torch.float32 or img.dim() != 4 or img.shape[1] != 3 or not img.is_contiguous() or mean.numel() != 3 or std.numel() != 3 or \
            mean.dtype != torch.float32 or std.dtype != torch.float32:
        raise HipError("spy_prep: contiguous fp32 (n,3,h,w) image and fp32 mean / std of 3 elements expected")
    n, _, h, w = img.shape
    out = torch.empty((n, h, w, 8), dtype=dtype, device=img.device)
    hip.check(hip.lib().vmg_spy_prep(hip.dtype_code(dtype), img.data_ptr(), mean.contiguous().data_ptr(), std.contiguous().data_ptr(), out.data_ptr(), n, h, w, hip.stream_ptr()),
              "vmg_spy_prep")
    return out


def spy_operand(ref: torch.Tensor, warped: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """[ref RGB | warped RGB | flow] (n,h,w,8) in ref's dtype from ref, warped (n,h,w,8: RGB in channels 0..2) and up (n,h,w,2) fp32 (vmg_spy_operand_fwd)."""
    hip.require_cuda(ref, warped, up)
    if ref.shape[-1] != 8 or warped.shape != ref.shape or warped.dtype != ref.dtype or up.dtype != torch.float32 or tuple(up.shape) != (*ref.shape[:-1], 2) or \
            not (ref.is_contiguous() and warped.is_contiguous() and up.is_contiguous()):
        raise HipError("spy_operand: contiguous (..., 8) images of one dtype and a contiguous fp32 (..., 2) flow expected")
    out = torch.empty_like(ref)
    hip.check(hip.lib().vmg_spy_operand_fwd(hip.dtype_code(ref.dtype), ref.data_ptr(), warped.data_ptr(), up.data_ptr(), out.data_ptr(), ref.numel() // 8,
                                            hip.stream_ptr()), "vmg_spy_operand_fwd")
    return out


def spy_operand_backward(dx8: torch.Tensor):
    """gradient of spy_operand's output -> (d warped (…, 8): channels 3..5 of dx8 in front of zeros, d up (…, 2) fp32: channels 6, 7)."""
    hip.require_cuda(dx8)
    if dx8.shape[-1] != 8 or not dx8.is_contiguous():
        raise HipError("spy_operand_backward: contiguous (..., 8) gradient expected")
    dwarped = torch.empty_like(dx8)
    dup = torch.empty((*dx8.shape[:-1], 2), dtype=torch.float32, device=dx8.device)
    hip.check(hip.lib().vmg_spy_operand_bwd(hip.dtype_code(dx8.dtype), dx8.data_ptr(), dwarped.data_ptr(), dup.data_ptr(), dx8.numel() // 8, hip.stream_ptr()),
              "vmg_spy_operand_bwd")
    return dwarped, dup


def spy_flow_add(up: torch.Tensor, res: torch.Tensor) -> torch.Tensor:
    """up (fp32) + res (any compute dtype) -> fp32, one pass (vmg_spy_flow_add)."""
    hip.require_cuda(up, res)
    if up.dtype != torch.float32 or up.shape != res.shape or not (up.is_contiguous() and res.is_contiguous()):
        raise HipError("spy_flow_add: contiguous tensors of one shape, the first fp32")
    out = torch.empty_like(up)
    hip.check(hip.lib().vmg_spy_flow_add(hip.dtype_code(res.dtype), up.data_ptr(), res.data_ptr(), out.data_ptr(), up.numel(), hip.stream_ptr()), "vmg_spy_flow_add")
    return out


def upsample2x_ac(x: torch.Tensor, scale: float, backward: bool = False) -> torch.Tensor:
    """forward: (n,h,w,c) fp32 -> scale * bilinear x2 (align_corners=True) (n,2h,2w,c); backward: the transpose on (n,2h,2w,c)."""
    hip.require_cuda(x)
    if x.dtype != torch.float32 or not x.is_contiguous() or x.dim() != 4:
        raise HipError("upsample2x_ac: contiguous fp32 (n,h,w,c) expected")
    n, h, w, c = x.shape
    if not backward:
        y = torch.empty((n, 2 * h, 2 * w, c), dtype=torch.float32, device=x.device)
        hip.check(hip.lib().vmg_upsample2x_ac_fwd(x.data_ptr(), y.data_ptr(), n, h, w, c, scale, hip.stream_ptr()), "vmg_upsample2x_ac_fwd")
        return y
    if h % 2 or w % 2:
        raise HipError("upsample2x_ac backward: even gradient size expected")
    dx = torch.empty((n, h // 2, w // 2, c), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().vmg_upsample2x_ac_bwd(x.data_ptr(), dx.data_ptr(), n, h // 2, w // 2, c, scale, hip.stream_ptr()), "vmg_upsample2x_ac_bwd")
    return dx


def space_depth_ln_forward(x: torch.Tensor, mode: str, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5):
    """UpdownkeepSampling's rearrangement + LayerNorm in one pass.  x: contiguous (N, Hin, Win, Cin); 'down' -> rows (N, Hin/2, Win/2, 4*Cin),
    'up' -> rows (N, 2*Hin, 2*Win, Cin/4).  Returns (y, mean, rstd)."""
    hip.require_cuda(x, w, b)
    N, Hi, Wi, Ci = x.shape
    if not x.is_contiguous() or w.dtype != torch.float32 or b.dtype != torch.float32:
        raise HipError("space_depth_ln: contiguous x, fp32 w / b expected")
    if mode == "down":
        if Hi % 2 or Wi % 2:
            raise HipError("space_depth_ln (down): even input size expected")
        H, W, cseg, C, m = Hi // 2, Wi // 2, Ci, 4 * Ci, 1
    else:
        if Ci % 4:
            raise HipError("space_depth_ln (up): channels must be a multiple of 4")
        H, W, cseg, C, m = 2 * Hi, 2 * Wi, Ci // 4, Ci // 4, 2
    if w.numel() != C or b.numel() != C:
        raise HipError("space_depth_ln: w / b must have the LayerNorm width")
    y = torch.empty((N, H, W, C), dtype=x.dtype, device=x.device)
    mean = torch.empty(N * H * W, dtype=torch.float32, device=x.device)
    rstd = torch.empty(N * H * W, dtype=torch.float32, device=x.device)
    hip.check(hip.lib().vmg_space_depth_ln_fwd(hip.dtype_code(x.dtype), m, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                               rstd.data_ptr(), N, H, W, cseg, eps, hip.stream_ptr()), "vmg_space_depth_ln_fwd")
    return y, mean, rstd


def space_depth_ln_backward(dy: torch.Tensor, x: torch.Tensor, mode: str, mean: torch.Tensor, rstd: torch.Tensor, w: torch.Tensor, into=None):
    hip.require_cuda(dy, x, mean, rstd, w)
    N, Hi, Wi, Ci = x.shape
    if mode == "down":
        H, W, cseg, C, m = Hi // 2, Wi // 2, Ci, 4 * Ci, 1
    else:
        H, W, cseg, C, m = 2 * Hi, 2 * Wi, Ci // 4, Ci // 4, 2
    dy = dy.contiguous()
    dx = torch.empty_like(x)
    dw, db = _grad_pair(C, x.device, into)
    hip.check(hip.lib().vmg_space_depth_ln_bwd(hip.dtype_code(x.dtype), m, dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), w.data_ptr(),
                                               dx.data_ptr(), dw.data_ptr(), db.data_ptr(), N, H, W, cseg, hip.stream_ptr()), "vmg_space_depth_ln_bwd")
    return dx, dw, db


def conv_wgrad_batched(xs: Sequence[torch.Tensor], dys: Sequence[torch.Tensor], dW: torch.Tensor, db: Optional[torch.Tensor], ks: int,
                       N: int, H: int, W: int, scale: float = 1.0, o0: int = 0, i0: int = 0):
    """dW += scale * sum_p wgrad(xs[p], dys[p]) in as few launches as possible (16 pairs per launch)."""
    if len(xs) != len(dys) or not xs:
        raise HipError("conv_wgrad_batched: need equally many x and dy tensors")
    x0, d0 = xs[0], dys[0]
    hip.require_cuda(dW, db, *xs, *dys)
    if dW.dtype != torch.float32 or not dW.is_contiguous() or (db is not None and (db.dtype != torch.float32 or not db.is_contiguous())):
        raise HipError("parameter gradients must be contiguous fp32")
    M = N * H * W
    Cin, Cout = x0.shape[-1], d0.shape[-1]
    xps, dps = _pix_stride(x0), _pix_stride(d0)
    for x, d in zip(xs, dys):
        if x.dtype != x0.dtype or d.dtype != x0.dtype or x.shape[-1] != Cin or d.shape[-1] != Cout or x.numel() // Cin != M or \
                d.numel() // Cout != M or _pix_stride(x) != xps or _pix_stride(d) != dps:
            raise HipError("conv_wgrad_batched: all pairs must share shape, dtype and strides")
    O_total, I_total = dW.shape[0], dW.shape[1]
    kk = 1 if dW.dim() == 2 else dW.shape[2]
    if kk != ks or o0 + Cout > O_total or i0 + Cin > I_total or (db is not None and db.numel() != O_total):
        raise HipError(f"gradient tensor {tuple(dW.shape)} does not match conv (ks={ks}, Cout={Cout}+{o0}, Cin={Cin}+{i0})")
    code = hip.dtype_code(x0.dtype)
    l = hip.lib()
    ws = _wgrad_workspace(dW.device)
    for s in range(0, len(xs), 16):
        n = min(16, len(xs) - s)
        xa = (ctypes.c_void_p * n)(*[t.data_ptr() for t in xs[s:s + n]])
        da = (ctypes.c_void_p * n)(*[t.data_ptr() for t in dys[s:s + n]])
        hip.check(l.vmg_conv_wgrad_batched_ws(code, ks, n, xa, da, N, H, W, xps, Cin, dps, Cout, dW.data_ptr(), I_total, o0, i0,
                                              db.data_ptr() if db is not None else None, scale, ws.data_ptr(), ws.numel(),
                                              hip.stream_ptr()), "vmg_conv_wgrad_batched_ws")


def conv_wgrad3_multi_ok(x: torch.Tensor, dy: torch.Tensor, ks: int) -> bool:
    """Shapes vmg_conv_wgrad3_multi / vmg_linear_wgrad2_multi take: bf16, 3x3 or 1x1, pixel strides that are multiples of 8 channels, 16-byte
    aligned tensors (1x1: channel counts multiples of 8 and at least 2 048 pixels)."""
    if ks == 1 and (x.shape[-1] % 8 or dy.shape[-1] % 8 or x.numel() // x.shape[-1] < 2048):
        return False
    return (ks in (1, 3) and x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16 and x.stride(-1) == 1 and dy.stride(-1) == 1 and
            x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0)


def linear_wgrad2_multi(probs, M: int):
    """probs: list of (xs, dys, dW (O, I[, 1, 1]), db, scale) of ONE shape: dW_i += scale_i * sum_p dys_i[p]^T xs_i[p] over M pixels, eight
    problems per launch (vmg_linear_wgrad2_multi)."""
    xs0, dys0, dW0 = probs[0][:3]
    npairs = len(xs0)
    Cin, Cout = xs0[0].shape[-1], dys0[0].shape[-1]
    xps, dps = _pix_stride(xs0[0]), _pix_stride(dys0[0])
    I_total = dW0.shape[1]
    for xs, dys, dW, db, _ in probs:
        hip.require_cuda(dW, db, *xs, *dys)
        if len(xs) != npairs or len(dys) != npairs or dW.shape != dW0.shape or dW.dtype != torch.float32 or not dW.is_contiguous() or \
                (db is not None and (db.dtype != torch.float32 or not db.is_contiguous() or db.numel() != dW.shape[0])):
            raise HipError("linear_wgrad2_multi: problems must share the shape; gradients contiguous fp32")
        for x, d in zip(xs, dys):
            if x.dtype != torch.bfloat16 or d.dtype != torch.bfloat16 or x.shape[-1] != Cin or d.shape[-1] != Cout or x.numel() // Cin != M or \
                    d.numel() // Cout != M or _pix_stride(x) != xps or _pix_stride(d) != dps:
                raise HipError("linear_wgrad2_multi: all pairs must share shape, dtype and strides")
    if dW0.shape[0] != Cout or Cin != I_total or dW0.numel() != Cout * Cin:
        raise HipError("linear_wgrad2_multi: gradient tensor does not match the layer")
    l = hip.lib()
    ws = _wgrad_workspace(dW0.device)
    for s in range(0, len(probs), 8):
        grp = probs[s:s + 8]
        n = len(grp)
        xa = (ctypes.c_void_p * (n * npairs))(*[t.data_ptr() for g in grp for t in g[0]])
        da = (ctypes.c_void_p * (n * npairs))(*[t.data_ptr() for g in grp for t in g[1]])
        wa = (ctypes.c_void_p * n)(*[g[2].data_ptr() for g in grp])
        ba = (ctypes.c_void_p * n)(*[(g[3].data_ptr() if g[3] is not None else None) for g in grp])
        sa = (ctypes.c_float * n)(*[float(g[4]) for g in grp])
        hip.check(l.vmg_linear_wgrad2_multi(n, npairs, xa, da, M, xps, Cin, dps, Cout, wa, I_total, 0, 0, ba, sa, ws.data_ptr(), ws.numel(),
                                            hip.stream_ptr()), "vmg_linear_wgrad2_multi")


def conv_wgrad3_multi(probs, N: int, H: int, W: int):
    """probs: list of (xs, dys, dW, db, scale) of ONE shape (same channel counts, strides and number of pairs): dW_i += scale_i * sum_p
    wgrad(xs_i[p], dys_i[p]), eight problems per launch."""
    xs0, dys0, dW0 = probs[0][:3]
    npairs = len(xs0)
    Cin, Cout = xs0[0].shape[-1], dys0[0].shape[-1]
    xps, dps = _pix_stride(xs0[0]), _pix_stride(dys0[0])
    I_total = dW0.shape[1]
    M = N * H * W
    for xs, dys, dW, db, _ in probs:
        hip.require_cuda(dW, db, *xs, *dys)
        if len(xs) != npairs or len(dys) != npairs or dW.shape != dW0.shape or dW.dtype != torch.float32 or not dW.is_contiguous() or \
                (db is not None and (db.dtype != torch.float32 or not db.is_contiguous() or db.numel() != dW.shape[0])):
            raise HipError("conv_wgrad3_multi: problems must share the shape; gradients contiguous fp32")
        for x, d in zip(xs, dys):
            if x.dtype != torch.bfloat16 or d.dtype != torch.bfloat16 or x.shape[-1] != Cin or d.shape[-1] != Cout or x.numel() // Cin != M or \
                    d.numel() // Cout != M or _pix_stride(x) != xps or _pix_stride(d) != dps:
                raise HipError("conv_wgrad3_multi: all pairs must share shape, dtype and strides")
    if dW0.shape[0] != Cout or dW0.dim() != 4 or dW0.shape[2] != 3 or Cin > I_total:
        raise HipError("conv_wgrad3_multi: gradient tensor does not match the convolution")
    l = hip.lib()
    ws = _wgrad_workspace(dW0.device)
    for s in range(0, len(probs), 8):
        grp = probs[s:s + 8]
        n = len(grp)
        xa = (ctypes.c_void_p * (n * npairs))(*[t.data_ptr() for g in grp for t in g[0]])
        da = (ctypes.c_void_p * (n * npairs))(*[t.data_ptr() for g in grp for t in g[1]])
        wa = (ctypes.c_void_p * n)(*[g[2].data_ptr() for g in grp])
        ba = (ctypes.c_void_p * n)(*[(g[3].data_ptr() if g[3] is not None else None) for g in grp])
        sa = (ctypes.c_float * n)(*[float(g[4]) for g in grp])
        hip.check(l.vmg_conv_wgrad3_multi(n, npairs, xa, da, N, H, W, xps, Cin, dps, Cout, wa, I_total, 0, 0, ba, sa, ws.data_ptr(), ws.numel(),
                                          hip.stream_ptr()), "vmg_conv_wgrad3_multi")


_WGRAD_WS = {}


def _wgrad_workspace(device) -> torch.Tensor:
    """Slab workspace of the large-tile weight-gradient kernel: allocated once per device, reused (calls are stream-ordered)."""
    key = str(device)
    ws = _WGRAD_WS.get(key)
    if ws is None:
        ws = torch.empty(int(hip.lib().vmg_conv_wgrad_ws_bytes()), dtype=torch.uint8, device=device)
        _WGRAD_WS[key] = ws
    return ws


def _ptrs(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def warp_bilinear_forward(x: torch.Tensor, flow: torch.Tensor) -> torch.Tensor:
    """x (n,h,w,c) contiguous, flow (n,h,w,2) fp32 contiguous -> bilinear, border padding."""
    hip.require_cuda(x, flow)
    n, h, w, c = x.shape
    if not x.is_contiguous() or flow.dtype != torch.float32 or not flow.is_contiguous() or tuple(flow.shape) != (n, h, w, 2):
        raise HipError("warp: x must be contiguous (n,h,w,c), flow contiguous fp32 (n,h,w,2)")
    out = torch.empty_like(x)
    hip.check(hip.lib().vmg_warp_bilinear_fwd(hip.dtype_code(x.dtype), x.data_ptr(), flow.data_ptr(), out.data_ptr(), n, h, w, c,
                                              hip.stream_ptr()), "vmg_warp_bilinear_fwd")
    return out


def warp_bilinear_backward(x: torch.Tensor, flow: torch.Tensor, dy: torch.Tensor):
    """-> (dx in x's dtype, dflow fp32)."""
    n, h, w, c = x.shape
    dy = dy.contiguous()
    # fp32 sums for bf16 tensors too, rounded once: the accumulator comes from the pool of zeroed buffers and goes back cleared by the rounding pass
    pooled = x.dtype != torch.float32 and (n * h * w * c) % 4 == 0
    dx_acc = ACC_POOL.take((n, h, w, c), x.device) if pooled else torch.zeros((n, h, w, c), dtype=torch.float32, device=x.device)
    dflow = torch.empty((n, h, w, 2), dtype=torch.float32, device=x.device)  # every element is written
    hip.check(hip.lib().vmg_warp_bilinear_bwd(hip.dtype_code(x.dtype), x.data_ptr(), flow.data_ptr(), dy.data_ptr(), dx_acc.data_ptr(),
                                              dflow.data_ptr(), n, h, w, c, hip.stream_ptr()), "vmg_warp_bilinear_bwd")
    if pooled:
        dx = cast_clear(dx_acc, x.dtype)
        ACC_POOL.give(dx_acc)
        return dx, dflow
    return dx_acc.to(x.dtype), dflow


def flow_smooth(x: torch.Tensor, r: int, backward: bool = False) -> torch.Tensor:
    """x (..., H, W) contiguous fp32 planes: forward the r x r block mean of the reflect-padded plane spread back over the block; backward=True: x is the output
    gradient, the result the input gradient (vmg_flow_smooth)."""
    hip.require_cuda(x)
    if x.dtype != torch.float32 or not x.is_contiguous() or x.dim() < 2:
        raise HipError("flow_smooth: contiguous fp32 (..., H, W) expected")
    H, W = x.shape[-2:]
    out = torch.empty_like(x)
    hip.check(hip.lib().vmg_flow_smooth(x.data_ptr(), out.data_ptr(), x.numel() // (H * W), H, W, int(r), 1 if backward else 0, hip.stream_ptr()), "vmg_flow_smooth")
    return out


def warp_nearest_planes(loc: torch.Tensor, flow: torch.Tensor) -> torch.Tensor:
    """loc (n,k2,h,w) fp32 planes advected with nearest sampling / border padding."""
    hip.require_cuda(loc, flow)
    n, k2, h, w = loc.shape
    if loc.dtype != torch.float32 or flow.dtype != torch.float32 or not flow.is_contiguous() or tuple(flow.shape) != (n, h, w, 2):
        raise HipError("warp_nearest_planes: fp32 loc (n,k2,h,w) and flow (n,h,w,2) expected")
    loc = loc.contiguous()
    out = torch.empty_like(loc)
    hip.check(hip.lib().vmg_warp_nearest_planes(loc.data_ptr(), flow.data_ptr(), out.data_ptr(), n, k2, h, w, hip.stream_ptr()),
              "vmg_warp_nearest_planes")
    return out


def ltam_forward(q, keys, vals, loc, rpe, decay, heads, wh, ww, scale):
    hip.require_cuda(q, loc, rpe, decay, *keys, *vals)
    n, h, w, c = q.shape
    t = len(keys)
    ts = [q] + list(keys) + list(vals)
    if any(x.dtype != q.dtype or tuple(x.shape) != (n, h, w, c) or not x.is_contiguous() for x in ts):
        raise HipError("ltam: q / keys / vals must be contiguous (n,h,w,c) tensors of one dtype")
    if loc.dtype != torch.float32 or tuple(loc.shape) != (n, 2 * t, h, w) or not loc.is_contiguous():
        raise HipError(f"ltam: loc must be contiguous fp32 (n,2t,h,w), got {tuple(loc.shape)}")
    if rpe.dtype != torch.float32 or decay.dtype != torch.float32 or not rpe.is_contiguous():
        raise HipError("ltam: rpe / decay must be fp32")
    out = torch.empty_like(q)
    lse = torch.empty((n, h, w, heads), dtype=torch.float32, device=q.device)
    hip.check(hip.lib().vmg_ltam_fwd(hip.dtype_code(q.dtype), q.data_ptr(), _ptrs(keys), _ptrs(vals), loc.data_ptr(), rpe.data_ptr(),
                                     decay.data_ptr(), out.data_ptr(), lse.data_ptr(), n, h, w, c, heads, wh, ww, t, scale,
                                     hip.stream_ptr()), "vmg_ltam_fwd")
    return out, lse


def ltam_backward(q, keys, vals, loc, rpe, decay, out, lse, dout, heads, wh, ww, scale, dk_into=None, dv_into=None, drpe_into=None):
    """dq, dk[j], dv[j], drpe.  dk / dv are FP32 sums (q's shape) for every tensor dtype: the caller rounds them once.  dk_into / dv_into: per
    key-frame an existing fp32 accumulator to scatter into (the gradient of a frame that several calls attend to is summed by the kernel's
    atomics, see functional.grad_bank), or None for a fresh zeroed one.  drpe_into: an fp32 tensor of rpe's shape to ADD the table gradient into (the
    parameter's .grad in the deferred weight-gradient mode) instead of a fresh zeroed one."""
    n, h, w, c = q.shape
    t = len(keys)
    dout = dout.contiguous()
    dq = torch.empty_like(q)
    dk = list(dk_into) if dk_into is not None else [None] * t
    dv = list(dv_into) if dv_into is not None else [None] * t
    for a in dk + dv:
        if a is not None and (a.shape != q.shape or a.dtype != torch.float32 or not a.is_contiguous()):
            raise HipError("ltam_backward: accumulators must be contiguous fp32 tensors of q's shape")
    fresh = [i for i, a in enumerate(dk + dv) if a is None]
    if fresh:  # ONE zero-fill for all new accumulators
        acc = torch.zeros((len(fresh), n, h, w, c), dtype=torch.float32, device=q.device)
        for slot, i in enumerate(fresh):
            if i < t:
                dk[i] = acc[slot]
            else:
                dv[i - t] = acc[slot]
    if drpe_into is not None and (drpe_into.shape != rpe.shape or drpe_into.dtype != torch.float32 or not drpe_into.is_contiguous()):
        raise HipError("ltam_backward: drpe_into must be a contiguous fp32 tensor of the table's shape")
    drpe = drpe_into if drpe_into is not None else torch.zeros_like(rpe)
    hip.check(hip.lib().vmg_ltam_bwd(hip.dtype_code(q.dtype), q.data_ptr(), _ptrs(keys), _ptrs(vals), loc.data_ptr(), rpe.data_ptr(),
                                     decay.data_ptr(), out.data_ptr(), lse.data_ptr(), dout.data_ptr(), dq.data_ptr(), _ptrs(dk), _ptrs(dv),
                                     drpe.data_ptr(), n, h, w, c, heads, wh, ww, t, scale, hip.stream_ptr()), "vmg_ltam_bwd")
    return dq, dk, dv, drpe


OP_CA_FWD, OP_CA_BWD, OP_MIX_FWD, OP_MIX_BWD, OP_GATE_FWD, OP_GATE_BWD, OP_AFFINE2, OP_SCALE, OP_GATE_RES_FWD, OP_GATE_RES_BWD = range(10)


def group_reduce(a: torch.Tensor, G: int, b: Optional[torch.Tensor] = None, c3: Optional[torch.Tensor] = None, mode: int = 0,
                 scale: float = 1.0) -> torch.Tensor:
    """a (and b, c3): contiguous (G*R, C)-shaped data; returns fp32 (G, C): scale * sum_r of (a+b+c3) or of a*b."""
    hip.require_cuda(a, b, c3)
    C = a.shape[-1]
    rows = a.numel() // C
    if rows % G or not a.is_contiguous() or any(t is not None and (t.shape != a.shape or t.dtype != a.dtype or not t.is_contiguous()) for t in (b, c3)):
        raise HipError("group_reduce: contiguous tensors of one shape / dtype covering G groups expected")
    out = torch.empty((G, C), dtype=torch.float32, device=a.device)
    ws = _group_reduce_workspace(a.device)
    hip.check(hip.lib().vmg_group_reduce(hip.dtype_code(a.dtype), a.data_ptr(), b.data_ptr() if b is not None else None,
                                         c3.data_ptr() if c3 is not None else None, out.data_ptr(), G, rows // G, C, mode, scale,
                                         ws.data_ptr(), ws.numel(), hip.stream_ptr()), "vmg_group_reduce")
    return out


_GR_WS = {}


def _group_reduce_workspace(device) -> torch.Tensor:
    """Block-partials workspace of the pooled sums: allocated once per device, reused (the calls are stream-ordered: a call's second launch
    has read the partials before the next call's first launch writes them)."""
    key = str(device)
    ws = _GR_WS.get(key)
    if ws is None:
        ws = _GR_WS[key] = torch.empty(int(hip.lib().vmg_group_reduce_ws_bytes()), dtype=torch.uint8, device=device)
    return ws


def group_reduce3(a: torch.Tensor, b0: torch.Tensor, b1: torch.Tensor, b2: torch.Tensor, G: int, scale: float = 1.0) -> torch.Tensor:
    """fp32 (G, C, 3): scale * sum_r a * {b0, b1, b2} over the rows of each group, one pass over a."""
    hip.require_cuda(a, b0, b1, b2)
    C = a.shape[-1]
    rows = a.numel() // C
    if rows % G or any(t.shape != a.shape or t.dtype != a.dtype or not t.is_contiguous() for t in (a, b0, b1, b2)):
        raise HipError("group_reduce3: contiguous tensors of one shape / dtype covering G groups expected")
    out = torch.empty((G, C, 3), dtype=torch.float32, device=a.device)
    ws = _group_reduce_workspace(a.device)
    hip.check(hip.lib().vmg_group_reduce3(hip.dtype_code(a.dtype), a.data_ptr(), b0.data_ptr(), b1.data_ptr(), b2.data_ptr(), out.data_ptr(), G,
                                          rows // G, C, scale, ws.data_ptr(), ws.numel(), hip.stream_ptr()), "vmg_group_reduce3")
    return out


def tab_elementwise(op: int, p0, p1=None, p2=None, coef=None, add=None, s: float = 1.0, G: int = 1, nout: int = 1):
    hip.require_cuda(p0, p1, p2, coef, add)
    C = p0.shape[-1]
    rows = p0.numel() // C
    for t in (p0, p1, p2):
        if t is not None and (not t.is_contiguous() or t.shape != p0.shape or t.dtype != p0.dtype):
            raise HipError("tab_elementwise: contiguous operands of one shape / dtype expected")
    for t in (coef, add):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise HipError("tab_elementwise: coefficients must be contiguous fp32")
    outs = [torch.empty_like(p0) for _ in range(nout)]
    ptr = lambda t: t.data_ptr() if t is not None else None
    o = outs + [None] * (3 - nout)
    hip.check(hip.lib().vmg_tab_elementwise(hip.dtype_code(p0.dtype), op, ptr(p0), ptr(p1), ptr(p2), ptr(coef), ptr(add), s, ptr(o[0]),
                                            ptr(o[1]), ptr(o[2]), rows, rows // G, C, hip.stream_ptr()), "vmg_tab_elementwise")
    return outs[0] if nout == 1 else outs


def se_mlp_forward(m: torch.Tensor, w1: torch.Tensor, b1, w2: torch.Tensor, b2, act1: int, mode: int):
    """Squeeze-excite MLP on pooled rows m (G, C) fp32: returns (pre (G, Hd), out (G, Co)); mode 0 sigmoid, mode 1 softmax over triples."""
    hip.require_cuda(m, w1, w2)
    G, C = m.shape
    Hd, Co = w1.shape[0], w2.shape[0]
    for t in (m, w1, w2, b1, b2):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise HipError("se_mlp: contiguous fp32 tensors only")
    pre = torch.empty(G, Hd, dtype=torch.float32, device=m.device)
    out = torch.empty(G, Co, dtype=torch.float32, device=m.device)
    hip.check(hip.lib().vmg_se_mlp_fwd(m.data_ptr(), w1.data_ptr(), b1.data_ptr() if b1 is not None else None, w2.data_ptr(),
                                       b2.data_ptr() if b2 is not None else None, pre.data_ptr(), out.data_ptr(), G, C, Hd, Co, act1, mode,
                                       hip.stream_ptr()), "vmg_se_mlp_fwd")
    return pre, out


def se_mlp_backward(dout: torch.Tensor, out: torch.Tensor, m: torch.Tensor, pre: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, act1: int,
                    mode: int, dm_scale: float, into=None):
    """-> dm (G, C) * dm_scale, dw1, db1, dw2, db2 (fresh tensors); into = (gw1, gb1, gw2, gb2): the parameter gradients are ADDED to these
    contiguous fp32 buffers (param.grad) and returned as such."""
    hip.require_cuda(dout, out, m, pre, w1, w2)
    G, C = m.shape
    Hd, Co = w1.shape[0], w2.shape[0]
    for t in (dout, out, m, pre, w1, w2):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise HipError("se_mlp: contiguous fp32 tensors only")
    dev = m.device
    sizes = (G * C, G * (Co + Hd)) if into is not None else (G * C, G * (Co + Hd), Hd * C, Hd, Co * Hd, Co)
    buf = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
    o = 0
    parts = []
    for n in sizes:
        parts.append(buf[o:o + n])
        o += n
    if into is not None:
        dm, ws = parts
        dw1, db1, dw2, db2 = into
        for t, n in zip(into, (Hd * C, Hd, Co * Hd, Co)):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n or not t.is_cuda:
                raise HipError("se_mlp: gradient buffers must be contiguous fp32 of the parameters' sizes")
    else:
        dm, ws, dw1, db1, dw2, db2 = parts
    hip.check(hip.lib().vmg_se_mlp_bwd(dout.data_ptr(), out.data_ptr(), m.data_ptr(), pre.data_ptr(), w1.data_ptr(), w2.data_ptr(), dm.data_ptr(),
                                       dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), ws.data_ptr(), G, C, Hd, Co, act1, mode,
                                       float(dm_scale), 1 if into is not None else 0, hip.stream_ptr()), "vmg_se_mlp_bwd")
    if into is not None:
        return dm.view(G, C), dw1, db1, dw2, db2
    return dm.view(G, C), dw1.view(Hd, C), db1, dw2.view(Co, Hd), db2


class PackPlan:
    """One-launch repacking of many weights (vmg_pack_entry / vmg_pack_run): the entry array lives on the device and is rebuilt only
    when the set of packs (or a weight's / pack buffer's address) changes."""

    def __init__(self):
        self.sig, self.dev, self.n, self.blocks = None, None, 0, 0
        # superseded entry tables are never freed: a captured hipGraph (train.TrainStep.capture, infer.GraphedModel) has the table's ADDRESS
        # baked into its vmg_pack_run node, and a later plan rebuild (an eager call at another shape creates new packs) must not hand that
        # memory back to the allocator while the graph can still be replayed.  A table is ~100 B per pack; plans are rebuilt a handful of times.
        self.retired = []

    def run(self, packs: Sequence[PackedConv], reuse: bool = False):
        """reuse: the caller guarantees that `packs` is the list of the previous call unless it has reset self.sig to None."""
        l = hip.lib()
        if reuse and self.sig is not None:
            hip.check(l.vmg_pack_run(self.dev.data_ptr(), self.n, self.blocks, hip.stream_ptr()), "vmg_pack_run")
            return
        packs = [p for p in packs if p.call is not None]
        if not packs:
            return
        sig = tuple((p.call[0], p.buf.data_ptr()) for p in packs)
        if sig != self.sig:
            esz = l.vmg_pack_entry_bytes()
            host = (ctypes.c_char * (esz * len(packs)))()
            base = ctypes.addressof(host)
            blk = 0
            owner = []  # entry index of every block: the kernel's blocks look their entry up here (appended to the entry array)
            for i, p in enumerate(packs):
                wptr, O, I, o0, on, soff, sch, tf = p.call
                nb = l.vmg_pack_entry(base + i * esz, 1 if p.layout == "ws" else 0, hip.dtype_code(p.dtype), wptr, O, I, p.ks, o0, on, len(sch),
                                      _intarr(soff), _intarr(sch), tf, p.cout_tiles, p.buf.data_ptr(), blk)
                if nb <= 0:
                    hip.check(nb if nb < 0 else -1, "vmg_pack_entry")
                blk += nb
                owner.extend([i] * nb)
            if self.dev is not None:
                self.retired.append(self.dev)
            table = struct.pack("<%di" % len(owner), *owner)
            self.dev = torch.frombuffer(bytearray(bytes(host) + table), dtype=torch.uint8).to(packs[0].buf.device)
            self.sig, self.n, self.blocks = sig, len(packs), blk
        hip.check(l.vmg_pack_run(self.dev.data_ptr(), self.n, self.blocks, hip.stream_ptr()), "vmg_pack_run")


# ---- fp8 (e4m3, block-scaled) convolution: Q8 records, packed weights, the kernel (csrc/conv_fp8.hip) ------------------------------------
def q8_record_bytes(C: int) -> int:
    return (C + 31) // 32 * 32 + 16


def q8_quantize(x: torch.Tensor) -> torch.Tensor:
    """bf16 channels-last (..., C) -> Q8 records uint8 (..., q8_record_bytes(C)) (vmg_q8_quantize)."""
    hip.require_cuda(x)
    C = x.shape[-1]
    if x.dtype != torch.bfloat16 or x.stride(-1) != 1:
        raise HipError("q8_quantize: bf16 channels-last tensor expected")
    ps = _pix_stride(x)
    M = x.numel() // C
    out = torch.empty(tuple(x.shape[:-1]) + (q8_record_bytes(C),), dtype=torch.uint8, device=x.device)
    hip.check(hip.lib().vmg_q8_quantize(x.data_ptr(), ps, out.data_ptr(), M, C, hip.stream_ptr()), "vmg_q8_quantize")
    return out


def q8_dequantize(rec: torch.Tensor, C: int) -> torch.Tensor:
    """Records -> fp32 (..., C) with torch ops (tests and tools; the product path never dequantises)."""
    nb = (C + 31) // 32
    data = rec[..., :nb * 32].contiguous().view(torch.float8_e4m3fn).float().reshape(*rec.shape[:-1], nb, 32)
    scale = torch.exp2(rec[..., nb * 32:nb * 32 + nb].float() - 127.0)
    return (data * scale[..., None]).reshape(*rec.shape[:-1], nb * 32)[..., :C]


class PackedQ8:
    __slots__ = ("buf", "cout", "cin", "call")

    def __init__(self, buf, cout, cin, call):
        self.buf, self.cout, self.cin, self.call = buf, cout, cin, call


def pack_conv_weight_q8(w: torch.Tensor, transpose_flip: bool = False, out: Optional[torch.Tensor] = None) -> PackedQ8:
    """fp32 (O, I, 3, 3) -> the fp8 convolution's weight image (vmg_convq8_pack): e4m3 with one power-of-two scale per output channel."""
    hip.require_cuda(w)
    if w.dtype != torch.float32 or not w.is_contiguous() or w.dim() != 4 or w.shape[2] != 3 or w.shape[3] != 3:
        raise HipError("pack_conv_weight_q8 expects a contiguous fp32 (O, I, 3, 3) weight")
    O, I = w.shape[0], w.shape[1]
    cout, cin = (I, O) if transpose_flip else (O, I)
    nbytes = hip.lib().vmg_convq8_pack_bytes(cout, cin)
    if nbytes <= 0:
        raise HipError(f"vmg_convq8_pack_bytes: {hip.lib().vmg_last_error().decode()}")
    buf = out if (out is not None and out.numel() >= int(nbytes)) else torch.empty(int(nbytes), dtype=torch.uint8, device=w.device)
    hip.check(hip.lib().vmg_convq8_pack(w.data_ptr(), O, I, 1 if transpose_flip else 0, buf.data_ptr(), hip.stream_ptr()), "vmg_convq8_pack")
    return PackedQ8(buf, cout, cin, (w.data_ptr(), O, I, 1 if transpose_flip else 0))


def q8_eligible(cout: int, cin: int) -> bool:
    return cout == cin and cout in (144, 112)


_Q8_DESC = hip.ConvQ8Desc()


def conv_q8_forward(src: torch.Tensor, pw: PackedQ8, bias: Optional[torch.Tensor], N: int, H: int, W: int, act: int = hip.ACT_NONE,
                    slope: float = 0.0, alpha: float = 1.0, res: Optional[torch.Tensor] = None, want_bf16: bool = True, want_q8: bool = True,
                    out: Optional[torch.Tensor] = None):
    """[res +] alpha * act(conv3x3(src records) + bias) -> (bf16 (N,H,W,Cout) or None, records (N,H,W,rec) or None) (vmg_convq8_fwd)."""
    hip.require_cuda(src, bias, res, out)
    rec_in = q8_record_bytes(pw.cin)
    if src.dtype != torch.uint8 or not src.is_contiguous() or src.shape[-1] != rec_in or src.numel() != N * H * W * rec_in:
        raise HipError(f"conv_q8: contiguous uint8 records (N*H*W, {rec_in}) expected, got {tuple(src.shape)}")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != pw.cout or not bias.is_contiguous()):
        raise HipError("conv_q8: bias must be contiguous fp32 of length Cout")
    if res is not None and (res.dtype != torch.bfloat16 or res.shape[-1] != pw.cout or res.numel() // pw.cout != N * H * W):
        raise HipError("conv_q8: the residual must be a bf16 tensor of the output's shape")
    o = oq = None
    if want_bf16:
        o = out if out is not None else torch.empty((N, H, W, pw.cout), dtype=torch.bfloat16, device=src.device)
    if want_q8:
        oq = torch.empty((N, H, W, q8_record_bytes(pw.cout)), dtype=torch.uint8, device=src.device)
    d = _Q8_DESC
    d.N, d.H, d.W, d.Cin, d.Cout = N, H, W, pw.cin, pw.cout
    d.src, d.packed, d.bias = src.data_ptr(), pw.buf.data_ptr(), bias.data_ptr() if bias is not None else None
    d.out, d.out_ps = (o.data_ptr(), _pix_stride(o)) if o is not None else (None, 0)
    d.outq = oq.data_ptr() if oq is not None else None
    d.res, d.res_ps = (res.data_ptr(), _pix_stride(res)) if res is not None else (None, 0)
    d.act, d.slope, d.alpha = act, slope, alpha
    hip.check(hip.lib().vmg_convq8_fwd(ctypes.byref(d), hip.stream_ptr()), "vmg_convq8_fwd")
    return o, oq


def resblock_chain_forward_q8(y0: torch.Tensor, pw1, b1, pw2, b2, r_scaling: float, keep_t: bool):
    """The fp8 part of a residual chain as ONE C call (vmg_resblock_chain_fwd_q8): y0 (N,H,W,C) bf16 -> (ys, ts); ts entries are None when not kept."""
    hip.require_cuda(y0, *b1, *b2)
    N, H, W, C = y0.shape
    nblk = len(pw1)
    if y0.dtype != torch.bfloat16 or not y0.is_contiguous() or len(pw2) != nblk or any(p.cout != C or p.cin != C for p in list(pw1) + list(pw2)):
        raise HipError("resblock_chain_forward_q8: contiguous bf16 y0 and C -> C fp8 packs expected")
    for b in list(b1) + list(b2):
        if b.dtype != torch.float32 or b.numel() != C or not b.is_contiguous():
            raise HipError("resblock_chain_forward_q8: biases must be contiguous fp32 of length C")
    rec = q8_record_bytes(C)
    q = q8_quantize(y0)
    qb = torch.empty((N, H, W, rec), dtype=torch.uint8, device=y0.device)
    blk = torch.empty((nblk * (2 if keep_t else 1), N, H, W, C), dtype=torch.bfloat16, device=y0.device).unbind(0)
    ys = [y0] + list(blk[:nblk])
    ts = list(blk[nblk:]) if keep_t else [None] * nblk
    d = hip.ChainQ8Desc()
    d.N, d.H, d.W, d.C, d.nblk = N, H, W, C, nblk
    d.q0, d.qa, d.qb = q.data_ptr(), q.data_ptr(), qb.data_ptr()
    keep = [_parr([p.buf for p in pw1]), _parr(list(b1)), _parr([p.buf for p in pw2]), _parr(list(b2)), _parr(ys), _parr(ts) if keep_t else None]
    d.packed1, d.bias1, d.packed2, d.bias2, d.y = keep[:5]
    if keep_t:
        d.t = keep[5]
    d.r_scaling = r_scaling
    hip.check(hip.lib().vmg_resblock_chain_fwd_q8(ctypes.byref(d), hip.stream_ptr()), "vmg_resblock_chain_fwd_q8")
    return ys, ts
