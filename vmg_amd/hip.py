"""ctypes binding of libvmg_hip.so (the C-ABI declared in include/vmg_hip.h).

The product path has no CPU or eager fallback: if the library is missing or an entry point fails, this
module raises.  Build it with ``python -m vmg_amd.build`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_void_p

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VMG_HIP_LIB") or os.path.join(HERE, "libvmg_hip.so")  # VMG_HIP_LIB: the diagnostics build (tools/ only)

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_GELU = 0, 1, 2, 3


class HipError(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    _fields_ = [
        ("dtype", c_int), ("ks", c_int), ("cout_tiles", c_int),
        ("N", c_int), ("H", c_int), ("W", c_int), ("Cout", c_int),
        ("nsrc", c_int),
        ("src", c_void_p * 4), ("src_ps", c_int64 * 4), ("src_ch", c_int * 4),
        ("packed", c_void_p), ("bias", c_void_p),
        ("out", c_void_p), ("out_ps", c_int64), ("out_pre", c_void_p),
        ("res", c_void_p), ("res_ps", c_int64),
        ("aux", c_void_p), ("aux_ps", c_int64),
        ("act", c_int), ("slope", c_float), ("alpha", c_float),
        ("actgrad", c_int), ("pixel_shuffle", c_int), ("mt", c_int), ("deep", c_int),
    ]


class ChainDesc(ctypes.Structure):
    """vmg_chain_desc (include/vmg_hip.h)."""
    _fields_ = [
        ("dtype", c_int), ("N", c_int), ("H", c_int), ("W", c_int), ("C", c_int), ("nblk", c_int), ("nsrc", c_int),
        ("src", c_void_p * 4), ("src_ps", c_int64 * 4), ("src_ch", c_int * 4),
        ("packed0", c_void_p), ("bias0", c_void_p), ("slope0", c_float), ("cout_tiles0", c_int), ("deep0", c_int),
        ("packed1", POINTER(c_void_p)), ("bias1", POINTER(c_void_p)), ("packed2", POINTER(c_void_p)), ("bias2", POINTER(c_void_p)),
        ("r_scaling", c_float), ("cout_tiles", c_int), ("deep", c_int),
        ("y", POINTER(c_void_p)), ("t", POINTER(c_void_p)), ("g_y", POINTER(c_void_p)), ("g_t", POINTER(c_void_p)),
    ]


class ConvQ8Desc(ctypes.Structure):
    """vmg_convq8_desc (include/vmg_hip.h)."""
    _fields_ = [
        ("N", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int), ("Cout", c_int),
        ("src", c_void_p), ("packed", c_void_p), ("bias", c_void_p),
        ("out", c_void_p), ("out_ps", c_int64), ("outq", c_void_p),
        ("res", c_void_p), ("res_ps", c_int64),
        ("act", c_int), ("slope", c_float), ("alpha", c_float),
    ]


class ChainQ8Desc(ctypes.Structure):
    """vmg_chainq8_desc (include/vmg_hip.h)."""
    _fields_ = [
        ("N", c_int), ("H", c_int), ("W", c_int), ("C", c_int), ("nblk", c_int),
        ("q0", c_void_p), ("qa", c_void_p), ("qb", c_void_p),
        ("packed1", POINTER(c_void_p)), ("bias1", POINTER(c_void_p)), ("packed2", POINTER(c_void_p)), ("bias2", POINTER(c_void_p)),
        ("y", POINTER(c_void_p)), ("t", POINTER(c_void_p)), ("r_scaling", c_float),
    ]


_lib = None

# name -> (restype, argtypes); every symbol of include/vmg_hip.h must be listed (tests/test_abi.py checks)
SIGNATURES = {
    "vmg_last_error": (c_char_p, []),
    "vmg_version": (c_int, []),
    "vmg_max_lds_bytes": (c_int, []),
    "vmg_conv_pack_bytes": (c_int64, [c_int, c_int, c_int, c_int, POINTER(c_int), c_int]),
    "vmg_conv_pack": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int),
                              c_int, c_int, c_void_p, c_void_p]),
    "vmg_convws_pack_bytes": (c_int64, [c_int, c_int, POINTER(c_int), c_int]),
    "vmg_convws_pack": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), c_int, c_int, c_void_p, c_void_p]),
    "vmg_conv_fwd": (c_int, [POINTER(ConvDesc), c_void_p]),
    "vmg_resblock_chain_fwd": (c_int, [POINTER(ChainDesc), c_void_p]),
    "vmg_resblock_chain_bwd": (c_int, [POINTER(ChainDesc), c_void_p]),
    "vmg_act_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_float, c_void_p]),
    "vmg_pixel_shuffle": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vmg_sum_n": (c_int, [c_int, c_void_p, c_int, c_void_p, c_int64, c_void_p]),
    "vmg_cast_clear": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "vmg_pair_steps": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "vmg_frame_gather": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "vmg_pixel_unshuffle_actgrad": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p]),
    "vmg_layernorm_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "vmg_layernorm_bwd_add": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "vmg_layernorm_bwd_multi": (c_int, [c_int, c_int, POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "vmg_layernorm_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "vmg_morphfc_fwd": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float,
                                c_void_p]),
    "vmg_morphfc_token_rows": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "vmg_morph_token_rows": (c_int64, [c_int, c_int, c_int, c_int, c_int]),
    "vmg_morph_tokens_gather": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vmg_morph_tokens_scatter": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vmg_win3d_attn_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_int, c_int, c_int, c_void_p]),
    "vmg_win3d_attn_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vmg_win3d_attn_bwd_ws_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "vmg_conv_wgrad3_multi": (c_int, [c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int64, c_int, c_void_p, c_int, c_int, c_int,
                                      c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "vmg_conv_wgrad3_variant": (c_int, [c_int]),
    "vmg_win3d_variant": (c_int, [c_int]),
    "vmg_linear_wgrad2_multi": (c_int, [c_int, c_int, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int64, c_int, c_void_p, c_int, c_int, c_int,
                                        c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "vmg_replay_build": (c_void_p, [c_void_p, c_void_p, c_void_p]),
    "vmg_replay_kernel_info": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vmg_replay_run": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "vmg_replay_destroy": (None, [c_void_p]),
    "vmg_pack_entry_bytes": (c_int, []),
    "vmg_pack_entry": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p,
                               c_int]),
    "vmg_pack_run": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "vmg_group_reduce3": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, ctypes.c_float, c_void_p, c_int64, c_void_p]),
    "vmg_group_reduce_ws_bytes": (c_int64, []),
    "vmg_se_mlp_fwd": (c_int, [c_void_p] * 7 + [c_int] * 6 + [c_void_p]),
    "vmg_se_mlp_bwd": (c_int, [c_void_p] * 12 + [c_int] * 6 + [ctypes.c_float, c_int, c_void_p]),
    "vmg_maxpool_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vmg_maxpool_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vmg_avgpool2_nhwc": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vmg_upsample2x_ac_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "vmg_upsample2x_ac_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "vmg_spy_operand_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "vmg_spy_operand_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "vmg_spy_flow_add": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "vmg_spy_prep": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "vmg_space_depth_ln_fwd": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "vmg_space_depth_ln_bwd": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                       c_void_p]),
    "vmg_conv_wgrad_batched": (c_int, [c_int, c_int, c_int, POINTER(c_void_p), POINTER(c_void_p), c_int, c_int, c_int, c_int64, c_int,
                                       c_int64, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_float, c_void_p]),
    "vmg_warp_bilinear_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vmg_warp_bilinear_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vmg_warp_nearest_planes": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vmg_flow_smooth": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "vmg_ltam_fwd": (c_int, [c_int, c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "vmg_ltam_bwd": (c_int, [c_int, c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_int, c_int, c_int, c_int, c_int,
                             c_int, c_int, c_int, c_float, c_void_p]),
    "vmg_charbonnier_edge_blocks": (c_int, [c_int64, c_int, c_int]),
    "vmg_charbonnier_edge_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p]),
    "vmg_charbonnier_edge_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_float, c_float, c_void_p]),
    "vmg_adamw_flat": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_float, c_float, c_float, c_void_p]),
    "vmg_q8_record_bytes": (c_int, [c_int]),
    "vmg_q8_quantize": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p]),
    "vmg_convq8_pack_bytes": (c_int64, [c_int, c_int]),
    "vmg_convq8_pack": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vmg_convq8_fwd": (c_int, [POINTER(ConvQ8Desc), c_void_p]),
    "vmg_resblock_chain_fwd_q8": (c_int, [POINTER(ChainQ8Desc), c_void_p]),
    "vmg_grad_clip_ws_bytes": (c_int64, []),
    "vmg_grad_clip_norm": (c_int, [c_void_p, c_int64, c_float, c_void_p, c_void_p, c_void_p]),
    "vmg_tile_accumulate": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                    c_int, c_void_p]),
    "vmg_tile_finalize": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "vmg_conv_debug_stamps": (c_int, [c_void_p]),
    "vmg_create": (c_void_p, [c_int]),
    "vmg_destroy": (c_int, [c_void_p]),
    "vmg_ctx_device": (c_int, [c_void_p]),
    "vmg_prof_select_pixels": (c_int, [c_void_p, c_int64]),
    "vmg_prof_null_interval_us": (ctypes.c_double, [c_int, c_void_p]),
    "vmg_conv_wgrad_ws_bytes": (c_int64, []),
    "vmg_conv_wgrad_batched_ws": (c_int, [c_int, c_int, c_int, POINTER(c_void_p), POINTER(c_void_p), c_int, c_int, c_int, c_int64, c_int,
                                          c_int64, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_int64, c_void_p]),
    "vmg_group_reduce": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_float, c_void_p, c_int64, c_void_p]),
    "vmg_tab_elementwise": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                    c_int64, c_int64, c_int, c_void_p]),
    "vmg_prof_begin": (c_int, [c_void_p, c_int, c_int, c_int]),
    "vmg_prof_end": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int), POINTER(ctypes.c_double)]),
    "vmg_conv_wgrad": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int,
                               c_void_p, c_int, c_int, c_int, c_void_p, c_float, c_void_p]),
}


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipError(f"{LIB_PATH} is missing: the VMG hot path has no fallback. Run `python -m vmg_amd.build`.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


_ctx = {}


def ctx(device: int = None) -> int:
    """The per-device handle (vmg_create), created on first use and kept for the life of the process."""
    if device is None:
        device = _cur_device() if _cur_device is not None else torch.cuda.current_device()
    h = _ctx.get(device)
    if h is None:
        h = lib().vmg_create(int(device))
        if not h:
            raise HipError(f"vmg_create({device}) failed: {lib().vmg_last_error().decode()}")
        _ctx[device] = h
    return h


def check(rc: int, what: str):
    if rc != 0:
        raise HipError(f"{what} failed ({rc}): {lib().vmg_last_error().decode()}")


def dtype_code(t: torch.dtype) -> int:
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise HipError(f"unsupported activation dtype {t}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr() -> int:
    """hipStream_t of torch's current stream.  The raw accessors cost ~0.3 us; torch.cuda.current_stream() builds a Stream
    object (~10 us) and this is called once per kernel launch (8 ms per train step before)."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    """Every tensor must live on the CURRENT device: kernels are enqueued on the current device's stream, and a pointer
    of another GPU there is a memory fault, not a Python error (use torch.cuda.set_device / torch.cuda.device(...))."""
    cur = -1
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise HipError("the VMG HIP path needs device tensors (there is no CPU fallback)")
        if cur < 0:
            cur = _cur_device() if _cur_device is not None else torch.cuda.current_device()
        if t.device.index != cur:
            raise HipError(f"tensor on cuda:{t.device.index} but the current device is cuda:{cur}: call torch.cuda.set_device first")
