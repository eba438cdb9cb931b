"""torch.autograd.Function wrappers around the HIP kernels (forward and hand-written backward).

PyTorch is the plumbing here (tensor storage, streams, the autograd graph); every contraction below runs in
libvmg_hip.so.  Activations are channels-last (..., C); parameters stay fp32 in checkpoint layout and are packed
(and cast to the compute dtype) on demand, cached per parameter version.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import hip
from . import kernels as K
from .hip import HipError

# ---------------------------------------------------------------------------------------------------------
# packed-weight cache: (id(param), kind, dtype, slices) -> (version, PackedConv)
# ---------------------------------------------------------------------------------------------------------
class _Fn(torch.autograd.Function):
    """torch.autograd.Function whose apply() skips the functorch bookkeeping (setup_context probing, dead-wrapper unwrapping of every
    argument): ~4 us per node and ~300 nodes per step on a host-bound step.  functorch transforms (vmap, functional grad) are not
    supported through these nodes -- the kernels are not batched-tensor aware anyway."""

    @classmethod
    def apply(cls, *args):
        return super(torch.autograd.Function, cls).apply(*args)


_PACK_CACHE = {}


def _pad_to(n: int, m: int = 8) -> int:
    return (n + m - 1) // m * m


def choose_tiling(M: int, cout: int, ks: int, dtype: torch.dtype, src_ch: Optional[Sequence[int]] = None, pixel_shuffle: bool = False):
    """(cout_tiles, mt, deep) for a conv over M pixels.  Measured on MI355X (tools/bench_conv.py under rocprofv3, kernel
    durations): the 144-channel 3x3 convs run best on the K-split kernel (deep = 2) as 64-pixel x 48-channel workgroups
    (three cout blocks, no padded tile, 138 registers -> three workgroups per CU): 25 us at M = 32 768 (27 us with two
    blocks of 80; 30 us on the pixel-split kernel), 77 us at M = 114 688 (84; 96)."""
    if USE_WS and src_ch is not None and not pixel_shuffle:
        t = K.ws_eligible(cout, ks, dtype, src_ch)
        if t and M <= 16384 and cout in (112, 144):
            # the recurrence of the full config (one clip per GPU: M = 2 * 64 * 64 = 8 192 pixels) has 64 of the weight-streaming kernel's 128-pixel
            # tiles for 256 CUs; the K-split kernel's 64-pixel x 64- / 48-channel workgroups fill the chip: 9.1 vs 11.4 us (112 channels), 10.9 vs
            # 14.2 us (144) at M = 8 192, 11.0 vs 11.4 / 13.2 vs 14.5 us at M = 16 384 (tools/bench_small_conv.py)
            return (4 if cout == 112 else 3), 1, 2
        if t:
            return t, 1, 3  # weight-streaming kernel: 128-pixel x 144/112-channel workgroups, the CU pulls the weights once
    if dtype == torch.bfloat16 and ks == 3 and cout in (144, 288):
        return 3, 1, 2  # 288 (local_cnn): 151 us vs 247 us at 144 -> 288, M = 114 688; the 576-channel PixelShuffle convs were slower this way
    if dtype == torch.bfloat16 and ks == 1 and src_ch is not None and len(src_ch) == 1 and M >= 65536 and not pixel_shuffle:
        if cout == 144 and src_ch[0] == 144:
            return 5, 1, 4  # the 144 -> 144 Linears of stage 0 (token mixers, proj): wave-autonomous kernel, 21 us vs 33 us at M = 114 688 (no gain at M = 32 768)
        if cout == 288 and src_ch[0] == 144:
            return 5, 1, 4  # Mlp_cnn.fc2's data gradient: 39 us vs 66 us (tools/bench_linear.py)
        if cout == 144 and src_ch[0] == 288:
            return 3, 1, 4  # Mlp_cnn.fc2 (the 288-channel source as two blocks of the pack): 48 us vs 55 us
    if dtype == torch.bfloat16 and ks == 3 and not pixel_shuffle and M >= (1 << 18) and src_ch is not None and len(src_ch) == 1 and \
            src_ch[0] <= 64 and src_ch[0] % 8 == 0 and cout in (16, 48, 64):  # (tile counts the general kernel has too)
        # the HR head (HRconv 64 -> 64, conv_last's data gradient 8 -> 64; 1.8 M pixels): weights-stationary kernel -- one workgroup per CU keeps
        # the layer's <= 74 KiB of weights in LDS and walks over 128-pixel tiles (the general kernel re-streams them per 64-pixel workgroup)
        return cout // 16, 1, 6
    if dtype == torch.bfloat16 and ks == 3 and not pixel_shuffle and M >= (1 << 20) and src_ch is not None and len(src_ch) == 1 and \
            (cout <= 16 or src_ch[0] <= 16):
        return None, 2, 0  # conv_last (64 -> 3) and its data gradient (8 -> 64) on 1.8 M pixels: 18 MFMAs per wave and tile -- 128-pixel tiles halve the workgroup count
    return None, 1, 0


USE_WS = True  # route eligible bf16 3x3 convs to the weight-streaming kernel (tests flip it to compare both)
GROUPED_DENSE = True  # bf16 grouped 3x3 convolutions as ONE launch on the dense block-diagonal pack (tests / tools flip it to compare both)

_WEIGHT_EPOCH = [0]


def bump_weight_epoch():
    """Invalidates every cached weight pack: called by optimizers that rewrite parameter memory outside autograd's
    version counters (train.FlatAdamW)."""
    _WEIGHT_EPOCH[0] += 1


def packed(weight: torch.Tensor, dtype: torch.dtype, kind: str, src_ch: Optional[Sequence[int]] = None,
           i0: int = 0, on: Optional[int] = None, tiles: Optional[int] = None, deep: int = 0, orange: Optional[Sequence[int]] = None,
           groups: int = 1) -> K.PackedConv:
    """kind 'fwd': outputs = all O, K slices = src_ch over I (padded to multiples of 8 with zero channels when
    needed).  kind 'dgrad': outputs = I[i0:i0+on), K = all O (padded to a multiple of 8).
    orange = (o0, no): one GROUP of a grouped convolution -- forward: only the outputs O[o0:o0+no); data gradient: K = O[o0:o0+no) (no % 8 == 0)."""
    if groups > 1:
        return _packed_grouped_dense(weight, dtype, kind, tiles, deep, groups)
    key = (id(weight), kind, dtype, tuple(src_ch) if src_ch else None, i0, on, tiles, deep == 3, tuple(orange) if orange else None)
    ver = (weight._version, _WEIGHT_EPOCH[0])
    hit = _PACK_CACHE.get(key)
    if hit is not None and hit[0] == ver and hit[2] is weight:
        return hit[1]
    if hit is not None and hit[0][1] == ver[1] and key not in _PACK_VOLATILE:
        _PACK_VOLATILE.add(key)  # modified in place between optimizer steps (the MorphFC decay, T1): repack_all leaves it alone
        _PACK_STAMP[0] += 1
    # a stale pack of this very parameter is REWRITTEN IN PLACE (stream order protects its earlier readers): its buffer's address may be baked
    # into a captured hipGraph (the pack node and the convolutions that read it), so it must neither move nor return to the allocator while
    # the parameter lives -- an eager call between replays used to re-create it and free the buffer the graph still wrote through
    buf0 = hit[1].buf if (hit is not None and hit[2] is weight) else None
    w = weight.detach()
    if w.dim() == 2:
        w = w[:, :, None, None]
    O, I = w.shape[0], w.shape[1]
    if kind == "fwd":
        if src_ch is None:
            src_ch = [I]
        if sum(src_ch) != I:
            raise HipError(f"source channels {list(src_ch)} do not sum to weight input channels {I}")
        if any(c % 8 for c in src_ch):
            # zero-pad each slice of the K dimension to a multiple of 8 channels
            parts, off = [], 0
            for c in src_ch:
                parts.append((off, c))
                off += c
            # (slice assignment into one zero tensor, not torch.cat: cat of contiguous pieces is a device-to-device memcpy, which a
            #  captured step cannot hand to csrc/replay.hip -- hipGraphMemcpyNodeGetParams does not describe 1-D copy nodes)
            wp = w.new_zeros(O, sum(_pad_to(c) for _, c in parts), *w.shape[2:])
            po = 0
            for o_, c in parts:
                wp[:, po:po + c] = w[:, o_:o_ + c]
                po += _pad_to(c)
            w = wp
            src_ch = [_pad_to(c) for c in src_ch]
        o0, no = (orange[0], orange[1]) if orange else (0, None)
        if deep == 3:
            pw = K.pack_conv_weight_ws(w.contiguous(), src_ch=list(src_ch), cout_tiles=tiles, o0=o0, on=no, out=buf0)
        else:
            pw = K.pack_conv_weight(w.contiguous(), dtype, src_ch=list(src_ch), cout_tiles=tiles, o0=o0, on=no, out=buf0)
    elif kind == "dgrad":
        on = I - i0 if on is None else on
        if orange:
            if orange[1] % 8:
                raise HipError("grouped data-gradient pack: the group's output channel count must be a multiple of 8")
            if deep == 3:
                pw = K.pack_conv_weight_ws(w.contiguous(), o0=i0, on=on, transpose_flip=True, cout_tiles=tiles, src_off=[orange[0]], src_ch=[orange[1]], out=buf0)
            else:
                pw = K.pack_conv_weight(w.contiguous(), dtype, o0=i0, on=on, transpose_flip=True, cout_tiles=tiles, src_off=[orange[0]], src_ch=[orange[1]], out=buf0)
            if key not in _PACK_VOLATILE and pw.call is not None and pw.call[0] == weight.data_ptr():
                _PACK_STAMP[0] += 1
            _PACK_CACHE[key] = [ver, pw, weight]
            return pw
        if O % 8:
            wp = w.new_zeros(_pad_to(O), *w.shape[1:])
            wp[:O].add_(w)  # (an add kernel into the zeros: a contiguous copy_ would be a memcpy node, see the forward pack)
            w = wp
        if deep == 3:
            pw = K.pack_conv_weight_ws(w.contiguous(), o0=i0, on=on, transpose_flip=True, cout_tiles=tiles, out=buf0)
        else:
            pw = K.pack_conv_weight(w.contiguous(), dtype, o0=i0, on=on, transpose_flip=True, cout_tiles=tiles, out=buf0)
    else:
        raise HipError(kind)
    if key not in _PACK_VOLATILE and pw.call is not None and pw.call[0] == weight.data_ptr():
        _PACK_STAMP[0] += 1  # (a new or re-created pack that repack_all serves: its plan is rebuilt; packs of zero-padded copies and of
        #                       weights modified in place are redone on demand every step and never enter the plan)
    _PACK_CACHE[key] = [ver, pw, weight]
    return pw


def _packed_grouped_dense(weight: torch.Tensor, dtype: torch.dtype, kind: str, tiles: Optional[int], deep: int, groups: int) -> K.PackedConv:
    """The dense block-diagonal pack of a grouped convolution's weight (O, cg, ks, ks): 'fwd' = all O outputs over groups * cg input channels,
    'dgrad' = all groups * cg outputs over K = O (multiples of 8).  Built from the parameter itself by the pack kernels (no dense copy), so it is
    served by the one-launch repack like any other pack."""
    key = (id(weight), kind, dtype, "dense", groups, tiles, deep == 3)
    ver = (weight._version, _WEIGHT_EPOCH[0])
    hit = _PACK_CACHE.get(key)
    if hit is not None and hit[0] == ver and hit[2] is weight:
        return hit[1]
    buf0 = hit[1].buf if (hit is not None and hit[2] is weight) else None
    w = weight.detach().contiguous()
    O, cg = w.shape[0], w.shape[1]
    if kind == "fwd":
        args = dict(src_ch=[cg * groups], cout_tiles=tiles, groups=groups, out=buf0)
    elif kind == "dgrad":
        if O % 8:
            raise HipError("grouped dense data-gradient pack: output channels must be a multiple of 8")
        args = dict(transpose_flip=True, cout_tiles=tiles, groups=groups, out=buf0)
    else:
        raise HipError(kind)
    pw = K.pack_conv_weight_ws(w, **args) if deep == 3 else K.pack_conv_weight(w, dtype, **args)
    if pw.call is not None and pw.call[0] == weight.data_ptr():
        _PACK_STAMP[0] += 1
    _PACK_CACHE[key] = [ver, pw, weight]
    return pw


def clear_pack_cache():
    _PACK_CACHE.clear()
    _PACK_VOLATILE.clear()
    _PACK_STAMP[0] += 1


_PACK_VOLATILE = set()
_PACK_PLAN = K.PackPlan()


_PACK_STAMP = [0]   # bumped when the set of plan-eligible packs changes (new key, a key turned volatile, cache cleared)
_PACK_STATE = {"stamp": -1, "ents": [], "packs": []}


def repack_all():
    """Rebuild, in ONE launch, every cached pack whose source is the parameter itself (call right after the optimizer has written the
    parameters): the next forward finds them current instead of launching ~390 small pack kernels one by one.  The list of packs is
    rebuilt only when the cache's membership has changed -- this runs at the very end of a step, with the GPU idle behind it."""
    st = _PACK_STATE
    if st["stamp"] != _PACK_STAMP[0]:
        ents, packs = [], []
        for key, ent in _PACK_CACHE.items():
            pw, weight = ent[1], ent[2]
            if key in _PACK_VOLATILE or pw.call is None or pw.call[0] != weight.data_ptr():
                continue  # (a zero-padded copy was packed, not the parameter: the on-demand path redoes it)
            ents.append(ent)
            packs.append(pw)
        fresh = True
        st["ents"], st["packs"], st["stamp"] = ents, packs, _PACK_STAMP[0]
    else:
        fresh = False
    if not st["packs"]:
        return
    _PACK_PLAN.run(st["packs"], reuse=not fresh)  # (fresh list: PackPlan compares addresses and re-uploads its entries only if they differ)
    ep = _WEIGHT_EPOCH[0]
    for ent in st["ents"]:
        ent[0] = (ent[2]._version, ep)


_VOL_PLAN = K.PackPlan()
_VOL_STATE = {"stamp": -1, "wkey": None, "ents": [], "packs": []}


def decay_weights_and_repack(weights: Sequence[torch.Tensor], gammas: Sequence[torch.Tensor]):
    """W <- W * Gamma for all the given weights in ONE launch and their cached packs rebuilt in ONE more (the MorphFC retention decay of every
    token mixer of a model, reference models/function.py:766-768 / 779-781, applied at the top of the model's forward instead of module by
    module: 24 small multiplies and 48 pack launches per step otherwise).  Packs that are not cached yet are built on demand as before."""
    with torch.no_grad():
        torch._foreach_mul_(list(weights), list(gammas))
    st = _VOL_STATE
    wkey = tuple(id(w) for w in weights)
    if st["stamp"] != _PACK_STAMP[0] or st["wkey"] != wkey:
        ids = set(wkey)
        ents = [ent for key, ent in _PACK_CACHE.items()
                if key in _PACK_VOLATILE and id(ent[2]) in ids and ent[1].call is not None and ent[1].call[0] == ent[2].data_ptr()]
        st["ents"], st["packs"], st["stamp"], st["wkey"] = ents, [e[1] for e in ents], _PACK_STAMP[0], wkey
        _VOL_PLAN.sig = None
        fresh = True
    else:
        fresh = False
    if not st["packs"]:
        return
    _VOL_PLAN.run(st["packs"], reuse=not fresh)
    ep = _WEIGHT_EPOCH[0]
    for ent in st["ents"]:
        ent[0] = (ent[2]._version, ep)


def _pad_channels(t: torch.Tensor, mult: int = 8) -> torch.Tensor:
    c = t.shape[-1]
    if c % mult == 0:
        return t
    base = getattr(t, "_vmg_padded", None)  # a channel slice of a tensor that already carries the zero channels (functional._MorphGather)
    if base is not None and base.shape[-1] == _pad_to(c, mult):
        return base
    return torch.nn.functional.pad(t, (0, _pad_to(c, mult) - c))


# ---------------------------------------------------------------------------------------------------------
# deferred, batched weight gradients
#
# The recurrence applies one conv module to every frame in both directions (2T uses per step).  A weight gradient
# per use has K = B*H*W pixels against a 144x144x9 fp32 output, so its float-atomic epilogue dominates.  Instead,
# backward only RECORDS (input, output-gradient) pairs; when the last use of a parameter has been seen the pairs
# are summed by ONE batched launch straight into param.grad (no zero-fill, no autograd accumulate kernels).
# 288 GB of HBM make keeping the pairs alive until then a non-issue.
# ---------------------------------------------------------------------------------------------------------
class _DeferredWgrad:
    """mode 'autograd' (default): every conv / Linear backward computes its weight gradient at once and returns it through
    autograd -- standard semantics, so torch DistributedDataParallel, GradScaler, clip_grad_norm_ and hooks all see it.
    mode 'deferred' (vmg_amd.train.TrainStep switches it on): backward only records the pairs, see above.

    Use counts are kept PER FORWARD PASS (a generation token taken in VMG.forward and stored in each autograd node), so
    a grad-enabled forward that is never back-propagated (an eval / logging call, a batch dropped after an exception)
    cannot leave counts behind that would silence a later step; and whatever is still pending when a backward() call ends
    is flushed by an end-of-backward engine callback, so no caller has to flush explicitly."""

    KEEP_GENERATIONS = 8

    def __init__(self):
        self.mode = "autograd"
        self.gen = 0
        self.uses = {}      # (generation, id(param)) -> outstanding forward uses
        self.pending = {}   # id(param) -> [weight, bias, entries]
        self.callbacks = []  # called with each parameter whose .grad has just been completed
        self.managed = set()  # ids of the parameters (weights and their biases) whose gradient is completed HERE, not by autograd
        self._queued = False
        self.ready = []     # [weight, bias, entries] whose last use has been seen, not batchable: launched by drain()
        self.waiting = {}   # multi-launch signature -> complete parameters waiting for company (launched at eight, or by drain())
        self.hold = 0       # > 0: a node that completes many parameters at once (a residual chain) is collecting them
        self.extra = {}     # (generation, id(param)) -> outstanding contributions of OTHER nodes to a managed bias (note_extra)
        self.held = {}      # id(param) -> param whose weight-gradient launch is done while such a contribution is still outstanding
        self.bw_gen = 0     # generation of the backward pass that is running (taken from the recorded entries)

    def begin_forward(self):
        """New top-level forward pass: a fresh generation; counts of passes older than KEEP_GENERATIONS are dropped."""
        self.gen += 1
        if self.uses or self.extra:
            lo = self.gen - self.KEEP_GENERATIONS
            for d in (self.uses, self.extra):
                for key in [k for k in d if k[0] < lo]:
                    del d[key]

    def note_use(self, weight, bias=None) -> int:
        key = (self.gen, id(weight))
        self.uses[key] = self.uses.get(key, 0) + 1
        self.managed.add(id(weight))
        if bias is not None:
            self.managed.add(id(bias))
        return self.gen

    # -- small parameters (LayerNorm affine, squeeze-excite MLPs): in mode 'deferred' their backward kernels add straight into .grad
    #    (no zero-filled temporaries, no AccumulateGrad add per parameter)
    def direct(self, *params) -> bool:
        return self.mode == "deferred" and all(p is not None and p.requires_grad and p.is_leaf for p in params)

    def note_params(self, *params) -> int:
        for p in params:
            key = (self.gen, id(p))
            self.uses[key] = self.uses.get(key, 0) + 1
            self.managed.add(id(p))
        return self.gen

    # -- a bias that a Linear / conv manages (its gradient is written by the deferred weight-gradient launch) may ALSO receive gradient from
    #    another node -- the 3-D window attention's q / kv biases, through the zero-padded positions (models/swin_3d.py: the padding is added
    #    before the Linears, so a padded token's q is the bias).  That node adds straight into .grad and the bias counts as complete only
    #    when BOTH have written: reporting it at the weight-gradient launch alone let the gradient reducer start the bucket's all-reduce
    #    while the attention backward's add was still to come (replicas diverge).
    def note_extra(self, *params) -> int:
        for p in params:
            key = (self.gen, id(p))
            self.extra[key] = self.extra.get(key, 0) + 1
            self.managed.add(id(p))
        return self.gen

    def extra_written(self, gen: int, *params):
        for p in params:
            key = (gen, id(p))
            left = self.extra.get(key, 1) - 1
            if left > 0:
                self.extra[key] = left
                continue
            self.extra.pop(key, None)
            if self.held.pop(id(p), None) is not None:
                for cb in self.callbacks:
                    cb(p)

    def _complete(self, p):
        """The deferred launch that writes p's gradient has been issued: report p, unless another node still owes it a contribution."""
        if self.extra and self.extra.get((self.bw_gen, id(p)), 0) > 0:
            self.held[id(p)] = p
            return
        for cb in self.callbacks:
            cb(p)

    @staticmethod
    def grad_of(p: torch.Tensor) -> torch.Tensor:
        if p.grad is None:
            p.grad = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
        return p.grad

    def written(self, gen: int, *params):
        for p in params:
            key = (gen, id(p))
            left = self.uses.get(key, 1) - 1
            if left <= 0:
                self.uses.pop(key, None)
                for cb in self.callbacks:
                    cb(p)
            else:
                self.uses[key] = left

    def add(self, weight, bias, srcs, src_ch, dpre, ks, N, H, W, scale: float = 1.0, gen: int = 0, o0: int = 0):
        """o0: first output channel of this use (a group of a grouped convolution writes rows o0 .. o0 + dpre channels of the gradient)."""
        ent = self.pending.setdefault(id(weight), [weight, bias, []])
        ent[2].append((srcs, tuple(src_ch), dpre, ks, N, H, W, float(scale), int(o0)))
        self.bw_gen = gen
        if not self._queued:  # whatever is still pending when this backward() call ends is completed then
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
            self._queued = True
        key = (gen, id(weight))
        left = self.uses.get(key, 1) - 1
        if left <= 0:
            self.uses.pop(key, None)
            self.flush(weight)
        else:
            self.uses[key] = left

    def _end_of_backward(self):
        self._queued = False
        for key in list(self.pending):
            self.flush(self.pending[key][0])
        self.hold = 0
        self.drain()
        if self.held:  # (a contribution that never came -- its node was not part of this backward: the gradient is what it is)
            held, self.held = self.held, {}
            for p in held.values():
                for cb in self.callbacks:
                    cb(p)

    def flush(self, weight):
        ent = self.pending.pop(id(weight), None)
        if ent is None:
            return
        sig = self._multi_sig(ent)  # computed ONCE per parameter and step (this runs on the autograd thread, with the GPU waiting behind it)
        if sig is None:
            if self.hold:
                self.ready.append(ent)
            else:
                self._launch([ent], None)
            return
        lst = self.waiting.setdefault(sig, [])
        lst.append(ent)
        if len(lst) >= 8 and not self.hold:
            self._launch(self.waiting.pop(sig), sig)

    @staticmethod
    def _multi_sig(ent):
        """Signature under which complete parameters can share one vmg_conv_wgrad3_multi / vmg_linear_wgrad2_multi launch, or None."""
        weight, _, entries = ent
        e0 = entries[0]
        ks = e0[3]
        if len(e0[1]) != 1 or ks not in (1, 3) or weight.shape[1] != e0[1][0] or (ks == 3 and weight.dim() != 4):
            return None
        if any(e[8] for e in entries) or e0[2].shape[-1] != weight.shape[0]:
            return None  # (groups of a grouped convolution: the general batched kernel, per output-row range)
        x0, d0 = e0[0][0], e0[2]
        if x0.shape[-1] != e0[1][0] or not K.conv_wgrad3_multi_ok(x0, d0, ks):
            return None
        xs0, ds0, xt0, dt0, c0, m0 = x0.shape, d0.shape, x0.stride(), d0.stride(), e0[1], e0[3:8]
        for e in entries[1:]:
            x, d = e[0][0], e[2]
            if e[1] != c0 or e[3:8] != m0 or x.shape != xs0 or d.shape != ds0 or x.stride() != xt0 or d.stride() != dt0 or not K.conv_wgrad3_multi_ok(x, d, ks):
                return None
        return (tuple(weight.shape), len(entries), e0[1], e0[4], e0[5], e0[6], tuple(xt0), tuple(dt0), ks)

    def _launch(self, ents, sig):
        """The gradients of the complete parameters `ents`: one by one (sig None or a single parameter) or eight per launch."""
        for weight, bias, _ in ents:
            if weight.grad is None:
                weight.grad = torch.zeros_like(weight, dtype=torch.float32)
            if bias is not None and bias.requires_grad and bias.grad is None:
                bias.grad = torch.zeros_like(bias, dtype=torch.float32)
        if sig is None or len(ents) < 2:
            for weight, bias, entries in ents:
                _wgrad_entries(entries, weight.grad, bias.grad if (bias is not None and bias.requires_grad) else None)
        else:
            probs = [([e[0][0] for e in entries], [e[2] for e in entries], weight.grad,
                      bias.grad if (bias is not None and bias.requires_grad) else None, entries[0][7]) for weight, bias, entries in ents]
            if sig[8] == 3:
                K.conv_wgrad3_multi(probs, sig[3], sig[4], sig[5])
            else:
                K.linear_wgrad2_multi(probs, sig[3] * sig[4] * sig[5])
        for weight, bias, _ in ents:
            for cb in self.callbacks:
                cb(weight)
            if bias is not None and bias.requires_grad:
                self._complete(bias)

    def drain(self):
        """Launch everything that is complete: parameters of one shape share launches (eight per launch).  Between drains (a residual
        chain completing, the end of the backward pass) shapes that can share a launch wait in `waiting` until eight of them are complete
        -- the two 3x3 convs of every RCAB, one per TAB, cost three launches per step instead of 24."""
        if self.ready:
            ready, self.ready = self.ready, []
            self._launch(ready, None)
        if self.waiting:
            waiting, self.waiting = self.waiting, {}
            for sig, ents in waiting.items():
                self._launch(ents, sig)

    def flush_all(self):
        self._end_of_backward()
        self.uses.clear()


def _wgrad_entries(entries, dW, db):
    """dW (+= ) the weight gradient of every recorded (sources, src_ch, dpre, ks, N, H, W, scale) entry, batched by shape."""
    groups = {}
    for e in entries:
        sig = (e[1], e[3], e[4], e[5], e[6], e[2].dtype, tuple(e[2].shape[-1:]), e[7], e[8] if len(e) > 8 else 0)
        groups.setdefault(sig, []).append(e)
    for (src_ch, ks, N, H, W, _, _, scale, o0), es in groups.items():
        off = 0
        for i, c in enumerate(src_ch):
            xs = [e[0][i][..., :c] if e[0][i].shape[-1] != c else e[0][i] for e in es]
            K.conv_wgrad_batched(xs, [e[2] for e in es], dW, db if i == 0 else None, ks, N, H, W, scale=scale, i0=off, o0=o0)
            off += c


def _wgrad_now(weight, bias_needed: bool, srcs, src_ch, dpre, ks, N, H, W, scale: float = 1.0):
    """(dW, db) of one use, as fresh fp32 tensors (mode 'autograd')."""
    dW = torch.zeros(weight.shape, dtype=torch.float32, device=weight.device)
    db = torch.zeros(weight.shape[0], dtype=torch.float32, device=weight.device) if bias_needed else None
    _wgrad_entries([(srcs, tuple(src_ch), dpre, ks, N, H, W, float(scale), 0)], dW, db)
    return dW, db


DEFERRED = _DeferredWgrad()


def set_wgrad_mode(mode: str):
    """'autograd' (default; weight gradients flow through autograd, DDP-compatible) or 'deferred' (batched per parameter,
    written straight into .grad; the mode of vmg_amd.train.TrainStep / GradBucketReducer)."""
    if mode not in ("autograd", "deferred"):
        raise HipError(f"wgrad mode {mode!r}: 'autograd' or 'deferred'")
    if mode != DEFERRED.mode:
        DEFERRED.flush_all()
        DEFERRED.managed.clear()
        DEFERRED.mode = mode


def flush_deferred_wgrads():
    """Completes every pending deferred gradient now (the end-of-backward callback does this by itself; kept for callers
    that read .grad from inside a backward hook)."""
    DEFERRED.flush_all()


def _act_grad(dy: torch.Tensor, y: Optional[torch.Tensor], pre: Optional[torch.Tensor], act: int, slope: float, alpha: float):
    """d(out)/d(pre) applied to dy for out = act(pre) * alpha."""
    if act == hip.ACT_NONE:
        return dy if alpha == 1.0 else dy * alpha
    return K.act_backward(dy, pre if act == hip.ACT_GELU else y, act, slope, alpha)


class _ActTok:
    """What a convolution with a fused activation tells its ONLY consumer, when that consumer is a convolution too (conv2d(..., fuse_src_act=True)): the
    consumer's data-gradient launch multiplies by the activation's derivative in its epilogue (the `aux` / `actgrad` operands the residual chains use) and sets
    `masked`; the producer's backward then takes the gradient it receives as the pre-activation gradient and skips its own derivative pass (vmg_act_bwd: one full
    read-modify-write of the tensor per convolution -- 66 launches per train step before round 4).  ReLU: the same bits (the mask is 0 / 1); leaky ReLU and GELU:
    one rounding instead of two."""
    __slots__ = ("act", "slope", "alpha", "pre", "masked")

    def __init__(self, act, slope, alpha):
        self.act, self.slope, self.alpha, self.pre, self.masked = act, slope, alpha, None, False


_ACTGRAD_CODE = {hip.ACT_RELU: 1, hip.ACT_LRELU: 2, hip.ACT_GELU: 3}


class _Conv2d(_Fn):
    """out = [res +] alpha * act(conv(cat(srcs)) + bias), optional PixelShuffle(2) store."""

    @staticmethod
    def forward(ctx, weight, bias, res, cfg, *srcs):
        ks, act, slope, alpha, pixel_shuffle, N, H, W, tok, src_tok = cfg
        cfg = cfg[:8]
        ctx.tok, ctx.src_tok = tok, src_tok
        dt = srcs[0].dtype
        src_ch = [s.shape[-1] for s in srcs]
        srcs_p = [_pad_channels(s) for s in srcs]
        tiles, mt, deep = choose_tiling(N * H * W, weight.shape[0], ks, dt, src_ch, pixel_shuffle)
        # a PixelShuffle conv the weight-streaming kernel covers (upconv1: 144 -> 576) runs there and is shuffled by a second, HBM-bound pass:
        # 250 + 55 us against 579 us on the kernel with the fused shuffle store (M = 114 688)
        ps_after = False
        if pixel_shuffle and USE_WS and res is None:
            t_ws = K.ws_eligible(weight.shape[0], ks, dt, [s.shape[-1] for s in srcs_p])
            if t_ws:
                ps_after, tiles, mt, deep = True, t_ws, 1, 3
        pw = packed(weight, dt, "fwd", src_ch, tiles=tiles, deep=deep)
        need_pre = act == hip.ACT_GELU and any(ctx.needs_input_grad)
        out, pre = K.conv_forward(srcs_p, pw, bias, N, H, W, act=act, slope=slope, alpha=alpha, res=res,
                                  pixel_shuffle=pixel_shuffle and not ps_after, want_pre=need_pre, mt=mt, deep=deep)
        if ps_after:
            out = K.pixel_shuffle(out, N, H, W)
        if tok is not None:
            tok.pre = pre  # (GELU: the consumer's epilogue needs the pre-activation)
        ctx.cfg = cfg
        ctx.src_ch = src_ch
        ctx.src_shapes = [tuple(t.shape) for t in srcs]
        ctx.res_shape = tuple(res.shape) if res is not None else None
        ctx.has_res = res is not None
        ctx.has_bias = bias is not None
        ctx.defer = DEFERRED.mode == "deferred" and isinstance(weight, torch.nn.Parameter) and ctx.needs_input_grad[0] and \
            (bias is None or isinstance(bias, torch.nn.Parameter))
        if ctx.defer:
            ctx.gen = DEFERRED.note_use(weight, bias)
            ctx.bias_ref = bias
        # relu / lrelu derivatives come from the sign of the output (taken before the residual is added, so keep
        # the sign information only when there is no residual; with a residual the activation is NONE on this path)
        if act in (hip.ACT_RELU, hip.ACT_LRELU) and res is not None:
            raise HipError("activation + residual in one epilogue is not differentiable from the output alone")
        ctx.save_for_backward(weight, out if act in (hip.ACT_RELU, hip.ACT_LRELU) else None, pre, *srcs_p)
        return out

    @staticmethod
    def backward(ctx, dy):
        ks, act, slope, alpha, pixel_shuffle, N, H, W = ctx.cfg
        weight, y, pre = ctx.saved_tensors[:3]
        srcs_p = ctx.saved_tensors[3:]
        dy = dy.contiguous()
        d_res = dy.reshape(ctx.res_shape) if ctx.has_res else None
        if ctx.tok is not None and ctx.tok.masked:
            ctx.tok.masked = False
            dpre = dy  # the consumer's data-gradient launch applied the derivative (see _ActTok)
        elif pixel_shuffle and act != hip.ACT_GELU and dy.shape[-1] % (8 if dy.dtype == torch.bfloat16 else 4) == 0:
            # depth-to-space undone on the gradient and the activation derivative taken from the HR output in one pass
            dpre = K.pixel_unshuffle_actgrad(dy, y if act != hip.ACT_NONE else None, N, H, W, act, slope, alpha)
        else:
            if pixel_shuffle:
                dy = K.pixel_unshuffle(dy, N, H, W)
                y = K.pixel_unshuffle(y, N, H, W) if y is not None else None
            dpre = _act_grad(dy, y, pre, act, slope, alpha)
        O = weight.shape[0]
        I = weight.shape[1]
        d_w = d_b = None
        d_srcs: List[Optional[torch.Tensor]] = []
        off = 0
        dpre_p = _pad_channels(dpre)
        for i, c in enumerate(ctx.src_ch):
            if ctx.needs_input_grad[4 + i]:
                tiles, mt, deep = choose_tiling(N * H * W, c, ks, dpre.dtype, [weight.shape[0]])
                pw = packed(weight, dpre.dtype, "dgrad", None, off, c, tiles=tiles, deep=deep)
                st = ctx.src_tok
                if st is not None:  # the source is a convolution's activated output and this is its only consumer: its derivative in this launch's epilogue
                    ref = st.pre if st.act == hip.ACT_GELU else srcs_p[i]
                    dx, _ = K.conv_forward([dpre_p], pw, None, N, H, W, mt=mt, deep=deep, aux=ref.reshape(N, H, W, c), actgrad=_ACTGRAD_CODE[st.act],
                                           slope=st.slope, alpha=st.alpha)
                    st.masked = True
                else:
                    dx, _ = K.conv_forward([dpre_p], pw, None, N, H, W, mt=mt, deep=deep)
                d_srcs.append(dx.reshape(ctx.src_shapes[i]))
            else:
                d_srcs.append(None)
            off += c
        # the weight gradient reads the zero-padded copy through a channel slice (pixel stride 8): 16-byte vector loads instead of O scalar ones
        dpre_w = dpre_p[..., :O] if dpre_p.shape[-1] != O else dpre
        if ctx.defer:
            DEFERRED.add(weight, ctx.bias_ref, list(srcs_p), ctx.src_ch, dpre_w, ks, N, H, W, gen=ctx.gen)
        elif ctx.needs_input_grad[0]:
            d_w, d_b = _wgrad_now(weight, ctx.has_bias and ctx.needs_input_grad[1], list(srcs_p), ctx.src_ch, dpre_w, ks, N, H, W)
            if weight.dim() == 2:
                d_w = d_w.reshape(weight.shape)
        elif ctx.has_bias and ctx.needs_input_grad[1]:
            d_b = dpre.float().reshape(-1, O).sum(0)
        return (d_w, d_b, d_res, None, *d_srcs)


def conv2d(srcs: Sequence[torch.Tensor], weight: torch.Tensor, bias: Optional[torch.Tensor], N: int, H: int, W: int,
           ks: int = 3, act: int = hip.ACT_NONE, slope: float = 0.0, alpha: float = 1.0, res: Optional[torch.Tensor] = None,
           pixel_shuffle: bool = False, fuse_src_act: bool = False) -> torch.Tensor:
    """Channels-last convolution / linear over N*H*W pixels; srcs are virtually concatenated along channels.
    fuse_src_act: the caller's promise that the (single) source is the activated output of another conv2d call and has NO other consumer -- the activation's
    derivative is then applied by this convolution's data-gradient launch (see _ActTok)."""
    grad = torch.is_grad_enabled()
    tok = src_tok = None
    if grad and act != hip.ACT_NONE and res is None and not pixel_shuffle:
        tok = _ActTok(act, float(slope), float(alpha))
    if grad and fuse_src_act and len(srcs) == 1 and srcs[0].requires_grad and srcs[0].shape[-1] % 8 == 0:
        src_tok = getattr(srcs[0], "_vmg_act_tok", None)
    cfg = (ks, act, float(slope), float(alpha), bool(pixel_shuffle), int(N), int(H), int(W), tok, src_tok)
    out = _Conv2d.apply(weight, bias, res, cfg, *srcs)
    if tok is not None:
        out._vmg_act_tok = tok
    return out


class _GroupedConv2d(_Fn):
    """act(conv(x, weight, groups) + bias): nn.Conv2d(C, O, ks, groups=G) on channels-last x (N,H,W,C) as G launches of the convolution kernel
    that all address the PARAMETER itself (output rows / K range of the group in the pack), write channel slices of ONE output tensor, and
    record their (input, output-gradient) pairs with the group's row offset -- no weight or activation slices go through autograd (round 2
    ran the groups as separate convolutions on sliced tensors: a pack kernel per group and call, a concatenation, four zero-filled slice
    gradients per tensor; reference: Mlp_cnn.fc1 with n_groups = 4, models/function.py:50-79)."""

    @staticmethod
    def _group_sources(x, N, H, W, G, cg):
        """per-group views of x for the grouped weight-gradient launch (channel counts padded to 8 through ONE padded copy when needed)"""
        cgp = _pad_to(cg)
        if cgp != cg:
            xp = torch.nn.functional.pad(x.reshape(N, H, W, G, cg), (0, cgp - cg))
            return [xp[..., g, :] for g in range(G)]
        xf = x.reshape(N, H, W, G * cg)
        return [xf[..., g * cg:(g + 1) * cg] for g in range(G)]

    @staticmethod
    def forward(ctx, x, weight, bias, cfg):
        G, ks, act, slope, N, H, W = cfg
        O, cg = weight.shape[0], weight.shape[1]
        og, dt, M = O // G, x.dtype, N * H * W
        ctx.x_shape = tuple(x.shape)
        x = x.contiguous()
        # Round 4: ALL groups as ONE launch on the dense block-diagonal pack (functional._packed_grouped_dense) -- G times the multiplies, a
        # G-th of the launches, and full-width channel vectors instead of 28- / 56-channel slices: at the full configuration's stage sizes (7 x 64^2
        # .. 7 x 8^2 pixels) the grouped launches were latency-, not FLOP-bound (profiles/r04_a_full_step_kernels.txt).  bf16, 3x3, C and O multiples of 8.
        # (measured, tools/bench_grouped_conv.py, forward + backward: 412 -> 305 us at 7 x 64^2 x 112 ch, 334 -> 308 at 32^2 x 224, 329 -> 252 at 16^2 x 224,
        #  but 361 -> 416 at 8^2 x 448, where the dense pack streams 4 x the weights for 448 pixels: up to 256 channels only)
        ctx.dense = GROUPED_DENSE and dt == torch.bfloat16 and ks == 3 and (G * cg) % 8 == 0 and O % 8 == 0 and G * cg <= 256
        if ctx.dense:
            C = G * cg
            tiles, mt, deep = choose_tiling(M, O, ks, dt, [C])
            pw = packed(weight, dt, "fwd", tiles=tiles, deep=deep, groups=G)
            need_pre = act == hip.ACT_GELU and any(ctx.needs_input_grad)
            out, pre = K.conv_forward([x.reshape(N, H, W, C)], pw, bias, N, H, W, act=act, slope=slope, want_pre=need_pre, mt=mt, deep=deep)
            ctx.cfg = cfg
            ctx.has_bias = bias is not None
            ctx.defer = DEFERRED.mode == "deferred" and isinstance(weight, torch.nn.Parameter) and ctx.needs_input_grad[1] and \
                (bias is None or isinstance(bias, torch.nn.Parameter))
            if ctx.defer:
                ctx.gen = DEFERRED.note_params(*([weight] + ([bias] if bias is not None else [])))
                ctx.bias_ref = bias
            ctx.srcs = None
            ctx.save_for_backward(weight, out if act in (hip.ACT_RELU, hip.ACT_LRELU) else None, pre, x)
            return out
        cgp = _pad_to(cg)
        if cgp != cg:  # groups whose channel count is no multiple of 8 (112 / 4 = 28): ONE padded copy (N,H,W,G,cgp), group g = a strided channel slice
            xp = torch.nn.functional.pad(x.reshape(N, H, W, G, cg), (0, cgp - cg))
            srcs = [xp[..., g, :] for g in range(G)]
        else:
            xf = x.reshape(N, H, W, G * cg)
            srcs = [xf[..., g * cg:(g + 1) * cg] for g in range(G)]
        out = torch.empty((N, H, W, O), dtype=dt, device=x.device)
        need_pre = act == hip.ACT_GELU and any(ctx.needs_input_grad)
        pre = torch.empty_like(out) if need_pre else None
        tiles, mt, deep = choose_tiling(M, og, ks, dt, [cgp])
        for g in range(G):
            pw = packed(weight, dt, "fwd", [cg], tiles=tiles, deep=deep, orange=(g * og, og))
            K.conv_forward([srcs[g]], pw, None if bias is None else bias[g * og:(g + 1) * og], N, H, W, act=act, slope=slope,
                           out=out[..., g * og:(g + 1) * og], out_pre=None if pre is None else pre[..., g * og:(g + 1) * og], mt=mt, deep=deep)
        ctx.cfg = cfg
        ctx.has_bias = bias is not None
        ctx.defer = DEFERRED.mode == "deferred" and isinstance(weight, torch.nn.Parameter) and ctx.needs_input_grad[1] and \
            (bias is None or isinstance(bias, torch.nn.Parameter))
        if ctx.defer:
            # one use; its gradient is written straight into .grad by ONE launch over the G groups in the backward (note_params / written)
            ctx.gen = DEFERRED.note_params(*([weight] + ([bias] if bias is not None else [])))
            ctx.bias_ref = bias
        ctx.srcs = srcs  # (views of x / of its padded copy: kept for the weight gradient)
        ctx.save_for_backward(weight, out if act in (hip.ACT_RELU, hip.ACT_LRELU) else None, pre, None)
        return out

    @staticmethod
    def backward(ctx, dy):
        G, ks, act, slope, N, H, W = ctx.cfg
        weight, y, pre, xs = ctx.saved_tensors
        O, cg = weight.shape[0], weight.shape[1]
        og, M = O // G, N * H * W
        srcs = ctx.srcs if not ctx.dense else _GroupedConv2d._group_sources(xs, N, H, W, G, cg)
        dpre = _act_grad(dy.contiguous(), y, pre, act, slope, 1.0)
        dt = dpre.dtype
        dx = None
        if ctx.needs_input_grad[0] and ctx.dense:
            tiles, mt, deep = choose_tiling(M, G * cg, ks, dt, [O])
            pw = packed(weight, dt, "dgrad", tiles=tiles, deep=deep, groups=G)
            dx, _ = K.conv_forward([dpre], pw, None, N, H, W, mt=mt, deep=deep)
        elif ctx.needs_input_grad[0]:
            dx = torch.empty((N, H, W, G * cg), dtype=dt, device=dy.device)
            tiles, mt, deep = choose_tiling(M, cg, ks, dt, [og])
            for g in range(G):
                pw = packed(weight, dt, "dgrad", None, 0, cg, tiles=tiles, deep=deep, orange=(g * og, og))
                K.conv_forward([dpre[..., g * og:(g + 1) * og]], pw, None, N, H, W, out=dx[..., g * cg:(g + 1) * cg], mt=mt, deep=deep)
        d_w = d_b = None
        if ctx.defer:
            bias = ctx.bias_ref
            dW = DEFERRED.grad_of(weight)
            db = DEFERRED.grad_of(bias) if (bias is not None and bias.requires_grad) else None
            dys = [dpre[..., g * og:(g + 1) * og] for g in range(G)]
            if ks == 3 and all(K.conv_wgrad3_multi_ok(srcs[g], dys[g], 3) for g in range(G)):
                # the G groups are G problems of one shape: ONE launch (and one ordered reduce) instead of G of each -- a group's gradient is a
                # contiguous row slice of the parameter's gradient
                probs = [([srcs[g][..., :cg] if srcs[g].shape[-1] != cg else srcs[g]], [dys[g]], dW[g * og:(g + 1) * og],
                          None if db is None else db[g * og:(g + 1) * og], 1.0) for g in range(G)]
                K.conv_wgrad3_multi(probs, N, H, W)
            else:
                _wgrad_entries([([srcs[g]], (cg,), dys[g], ks, N, H, W, 1.0, g * og) for g in range(G)], dW, db)
            DEFERRED.written(ctx.gen, *([weight] + ([bias] if bias is not None else [])))
        elif ctx.needs_input_grad[1]:
            d_w = torch.zeros(weight.shape, dtype=torch.float32, device=weight.device)
            d_b = torch.zeros(O, dtype=torch.float32, device=weight.device) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
            _wgrad_entries([([srcs[g]], (cg,), dpre[..., g * og:(g + 1) * og], ks, N, H, W, 1.0, g * og) for g in range(G)], d_w, d_b)
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            d_b = dpre.float().reshape(-1, O).sum(0)
        return (dx.reshape(ctx.x_shape) if dx is not None else None), d_w, d_b, None


def grouped_conv2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], groups: int, N: int, H: int, W: int, ks: int = 3,
                   act: int = hip.ACT_NONE, slope: float = 0.0) -> torch.Tensor:
    """nn.Conv2d(C, O, ks, padding=ks//2, groups=groups) + activation on channels-last x covering N*H*W pixels -> (N,H,W,O)."""
    if weight.shape[0] % groups or x.shape[-1] != groups * weight.shape[1] or (weight.shape[0] // groups) % 8:
        raise HipError("grouped_conv2d: channels must divide into the groups (output channels per group a multiple of 8)")
    return _GroupedConv2d.apply(x, weight, bias, (int(groups), int(ks), int(act), float(slope), int(N), int(H), int(W)))


# ---- fp8 chains: conv1 / conv2 of every residual block on the block-scaled fp8 kernel (csrc/conv_fp8.hip) ----------------------
FP8_CHAINS = False          # process-wide default (set_fp8_chains); VMG(fp8_chains=True) switches it on per model
FP8_STATS = {"chains": 0}   # forward chains that ran on the fp8 kernel (tests)
_Q8_CACHE = {}


def set_fp8_chains(on: bool):
    global FP8_CHAINS
    FP8_CHAINS = bool(on)


def packed_q8(weight: torch.Tensor) -> K.PackedQ8:
    """The fp8 weight image of a 3x3 convolution, cached per parameter version / optimizer epoch (rebuilt on demand: two small launches)."""
    ver = (weight._version, _WEIGHT_EPOCH[0])
    hit = _Q8_CACHE.get(id(weight))
    if hit is not None and hit[0] == ver and hit[2] is weight:
        return hit[1]
    pw = K.pack_conv_weight_q8(weight.detach().contiguous(), out=hit[1].buf if (hit is not None and hit[2] is weight) else None)
    _Q8_CACHE[id(weight)] = (ver, pw, weight)
    return pw


def _chain_forward_fp8(srcs, params, r_scaling, keep_t: bool):
    """ResidualBlocksWithInputConv forward with the 2 * nblk block convolutions in fp8: y_0 = lrelu(conv0(srcs)) on the bf16 kernel, quantised to
    records once; then per block  t_k = relu(conv1(q(y_k)))  and  y_{k+1} = y_k + r * conv2(q(t_k))  on vmg_convq8_fwd -- every convolution
    writes the records its successor reads, the skip path stays bf16.  keep_t: also write t_k as bf16 (the backward needs it)."""
    nblk = (len(params) - 2) // 4
    N, H, W = srcs[0].shape[0], srcs[0].shape[1], srcs[0].shape[2]
    dt, M = srcs[0].dtype, srcs[0].shape[0] * srcs[0].shape[1] * srcs[0].shape[2]
    src_ch = [t.shape[-1] for t in srcs]
    w0, b0 = params[0], params[1]
    C = w0.shape[0]
    t0_, m0_, d0_ = choose_tiling(M, C, 3, dt, src_ch)
    y0, _ = K.conv_forward(srcs, packed(w0, dt, "fwd", src_ch, tiles=t0_, deep=d0_), b0, N, H, W, act=hip.ACT_LRELU, slope=0.1, mt=m0_, deep=d0_)
    ys, ts = K.resblock_chain_forward_q8(y0, [packed_q8(params[2 + 4 * k]) for k in range(nblk)], [params[3 + 4 * k] for k in range(nblk)],
                                         [packed_q8(params[4 + 4 * k]) for k in range(nblk)], [params[5 + 4 * k] for k in range(nblk)], r_scaling, keep_t)
    FP8_STATS["chains"] += 1
    return ys, ts


class _ResidualChain(_Fn):
    """ResidualBlocksWithInputConv (models/trajectory.py:16-52, 165-221) as ONE autograd node:
        y0 = lrelu_0.1(conv0(cat(srcs)));  y_{k+1} = y_k + r * conv2_k(relu(conv1_k(y_k)))
    Forward is the same fused-epilogue conv launches as the generic path; the point is the BACKWARD, which runs the
    data-gradient chain with everything fused into conv epilogues -- relu mask (actgrad), r scaling (alpha) and the
    skip-path gradient (res) -- so there are no separate activation-backward or gradient-add kernels, and which only
    records (input, output-gradient) pairs for the deferred batched weight gradient."""

    @staticmethod
    def _run(srcs, params, r_scaling, own_output):
        nblk = (len(params) - 2) // 4
        N, H, W = srcs[0].shape[0], srcs[0].shape[1], srcs[0].shape[2]
        dt = srcs[0].dtype
        M = N * H * W
        src_ch = [t.shape[-1] for t in srcs]
        w0, b0 = params[0], params[1]
        C = w0.shape[0]
        t0_, _, d0_ = choose_tiling(M, C, 3, dt, src_ch)
        tiles, _, deep = choose_tiling(M, C, 3, dt, [C])
        pw1 = [packed(params[2 + 4 * k], dt, "fwd", [C], tiles=tiles, deep=deep) for k in range(nblk)]
        pw2 = [packed(params[4 + 4 * k], dt, "fwd", [C], tiles=tiles, deep=deep) for k in range(nblk)]
        return K.resblock_chain_forward(srcs, packed(w0, dt, "fwd", src_ch, tiles=t0_, deep=d0_), b0, 0.1, d0_, pw1,
                                        [params[3 + 4 * k] for k in range(nblk)], pw2, [params[5 + 4 * k] for k in range(nblk)], r_scaling, deep,
                                        own_output=own_output)

    @staticmethod
    def forward(ctx, r_scaling, nsrc, fp8, *args):
        recompute = nsrc < 0  # (encoded in the sign: activation recompute -- keep the sources only, run the chain again in the backward)
        nsrc = abs(nsrc)
        srcs = list(args[:nsrc])
        params = args[nsrc:]  # w0, b0, then (w1, b1, w2, b2) per block
        nblk = (len(params) - 2) // 4
        N, H, W = srcs[0].shape[0], srcs[0].shape[1], srcs[0].shape[2]
        dt = srcs[0].dtype
        M = N * H * W
        src_ch = [t.shape[-1] for t in srcs]
        w0, b0 = params[0], params[1]
        C = w0.shape[0]
        fp8 = bool(fp8) and dt == torch.bfloat16 and nblk > 0 and K.q8_eligible(C, C)
        ctx.fp8 = fp8
        if fp8:
            ys, ts = _chain_forward_fp8(srcs, params, r_scaling, keep_t=any(ctx.needs_input_grad) and not recompute)
        else:
            t0_, _, d0_ = choose_tiling(M, C, 3, dt, src_ch)
            tiles, _, deep = choose_tiling(M, C, 3, dt, [C])
            pw1 = [packed(params[2 + 4 * k], dt, "fwd", [C], tiles=tiles, deep=deep) for k in range(nblk)]
            pw2 = [packed(params[4 + 4 * k], dt, "fwd", [C], tiles=tiles, deep=deep) for k in range(nblk)]
            ys, ts = K.resblock_chain_forward(srcs, packed(w0, dt, "fwd", src_ch, tiles=t0_, deep=d0_), b0, 0.1, d0_, pw1,
                                              [params[3 + 4 * k] for k in range(nblk)], pw2, [params[5 + 4 * k] for k in range(nblk)], r_scaling, deep,
                                              own_output=recompute)
        ctx.recompute = recompute
        ctx.meta = (r_scaling, nsrc, nblk, N, H, W, src_ch, C)
        ctx.wgrad = any(ctx.needs_input_grad[3 + nsrc:])
        ctx.defer = ctx.wgrad and DEFERRED.mode == "deferred"
        if ctx.defer:
            for p, pb in zip(params[0::2], params[1::2]):
                ctx.gen = DEFERRED.note_use(p, pb)
        ctx.params = params
        if recompute:
            ctx.save_for_backward(*srcs)  # 2 * nblk intermediates per call are dropped here and rebuilt by _run() in the backward
        else:
            ctx.save_for_backward(*srcs, *ys[:max(nblk, 1)], *ts)  # the final output is not needed (with no blocks y_0 is the output)
        return ys[-1]

    @staticmethod
    def backward(ctx, g):
        r, nsrc, nblk, N, H, W, src_ch, C = ctx.meta
        params = ctx.params
        srcs = list(ctx.saved_tensors[:nsrc])
        if ctx.recompute:
            # the forward once more: +1/3 of the chain's FLOPs, -31 saved tensors per call
            ys, ts = _chain_forward_fp8(srcs, params, r, True) if ctx.fp8 else _ResidualChain._run(srcs, params, r, False)
            ys = ys[:max(nblk, 1)]
        else:
            ys = list(ctx.saved_tensors[nsrc:nsrc + max(nblk, 1)])  # y_0 .. y_{nblk-1}
            ts = list(ctx.saved_tensors[nsrc + max(nblk, 1):])      # t_0 .. t_{nblk-1}
        y0 = ys[0]
        dt = g.dtype
        M = N * H * W
        tiles, mt, deep = choose_tiling(M, C, 3, dt, [C])
        pg = [None] * len(params)  # parameter gradients returned through autograd (mode 'autograd')
        pd1 = [packed(params[2 + 4 * k], dt, "dgrad", None, 0, C, tiles=tiles, deep=deep) for k in range(nblk)]
        pd2 = [packed(params[4 + 4 * k], dt, "dgrad", None, 0, C, tiles=tiles, deep=deep) for k in range(nblk)]
        gys, gts = K.resblock_chain_backward(g, ts, pd1, pd2, r, deep)
        if ctx.defer:
            DEFERRED.hold += 1  # the chain's parameters complete together: their gradients share launches (drained below)
        for k in range(nblk - 1, -1, -1):
            w1, b1, w2, b2 = params[2 + 4 * k: 6 + 4 * k]
            if ctx.defer:
                DEFERRED.add(w2, b2, [ts[k]], [C], gys[k + 1], 3, N, H, W, scale=r, gen=ctx.gen)
                DEFERRED.add(w1, b1, [ys[k]], [C], gts[k], 3, N, H, W, gen=ctx.gen)
            elif ctx.wgrad:
                pg[4 + 4 * k], pg[5 + 4 * k] = _wgrad_now(w2, True, [ts[k]], [C], gys[k + 1], 3, N, H, W, scale=r)
                pg[2 + 4 * k], pg[3 + 4 * k] = _wgrad_now(w1, True, [ys[k]], [C], gts[k], 3, N, H, W)
        g = gys[0]
        w0, b0 = params[0], params[1]
        dpre0 = K.act_backward(g, y0, hip.ACT_LRELU, 0.1, 1.0)
        d_srcs = []
        off = 0
        for i, c in enumerate(src_ch):
            if ctx.needs_input_grad[3 + i]:
                t_, m_, d_ = choose_tiling(M, c, 3, dt, [C])
                dx, _ = K.conv_forward([dpre0], packed(w0, dt, "dgrad", None, off, c, tiles=t_, deep=d_), None, N, H, W, mt=m_, deep=d_)
                d_srcs.append(dx)
            else:
                d_srcs.append(None)
            off += c
        if ctx.defer:
            DEFERRED.add(w0, b0, srcs, src_ch, dpre0, 3, N, H, W, gen=ctx.gen)
            DEFERRED.hold -= 1
            if not DEFERRED.hold:
                DEFERRED.drain()
        elif ctx.wgrad:
            pg[0], pg[1] = _wgrad_now(w0, True, srcs, src_ch, dpre0, 3, N, H, W)
        return (None, None, None, *d_srcs, *pg)


def residual_chain(srcs: Sequence[torch.Tensor], conv0, blocks, r_scaling: float, recompute: bool = False, fp8: bool = False) -> torch.Tensor:
    """srcs: channels-last (n,h,w,c_s) tensors (virtual concat); conv0 and blocks[k].conv1/.conv2 are nn.Conv2d holders.
    recompute: keep only the sources for the backward and run the chain again there (activation recompute).
    fp8 (or functional.set_fp8_chains(True)): the block convolutions of the FORWARD run on the fp8 kernel (bf16 tensors, 144 / 112 channels)."""
    params = [conv0.weight, conv0.bias]
    for b in blocks:
        params += [b.conv1.weight, b.conv1.bias, b.conv2.weight, b.conv2.bias]
    ok = all(isinstance(p, torch.nn.Parameter) for p in params) and all(t.is_contiguous() and t.shape[-1] % 8 == 0 for t in srcs)
    if not ok:
        raise HipError("residual_chain needs contiguous sources with multiples of 8 channels and nn.Parameter weights")
    return _ResidualChain.apply(float(r_scaling), -len(srcs) if recompute else len(srcs), bool(fp8 or FP8_CHAINS), *srcs, *params)


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: int = hip.ACT_NONE, alpha: float = 1.0,
           res: Optional[torch.Tensor] = None, slope: float = 0.0, fuse_src_act: bool = False) -> torch.Tensor:
    """y[..., O] = act(x[..., I] @ W^T + b) * alpha (+ res): the KS = 1 convolution on (M, C) rows."""
    M = x.numel() // x.shape[-1]
    out = conv2d([x], weight, bias, 1, 1, M, ks=1, act=act, slope=slope, alpha=alpha,
                 res=None if res is None else res, fuse_src_act=fuse_src_act)
    y = out.reshape(*x.shape[:-1], weight.shape[0])
    tok = getattr(out, "_vmg_act_tok", None)
    if tok is not None and y is not out:
        y._vmg_act_tok = tok  # (the reshape is a view of the same values: the consumer may still fuse the derivative)
    return y


class _LayerNorm(_Fn):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        x = x.contiguous()
        y, mean, rstd = K.layernorm_forward(x, w, b, eps)
        ctx.save_for_backward(x, mean, rstd, w)
        ctx.direct = DEFERRED.direct(w, b)
        if ctx.direct:
            ctx.params, ctx.gen = (w, b), DEFERRED.note_params(w, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, w = ctx.saved_tensors
        if ctx.direct:
            dx, _, _ = K.layernorm_backward(dy, x, mean, rstd, w, into=tuple(DEFERRED.grad_of(p) for p in ctx.params))
            DEFERRED.written(ctx.gen, *ctx.params)
            return dx, None, None, None
        dx, dw, db = K.layernorm_backward(dy, x, mean, rstd, w)
        return dx, dw, db, None


class _SpaceDepthLayerNorm(_Fn):
    """UpdownkeepSampling's space<->depth rearrangement fused into its LayerNorm (models/layers.py:785-793): rows are gathered from
    the feature map in the forward and their gradient is scattered back in the backward -- no rearranged copy either way."""

    @staticmethod
    def forward(ctx, x, w, b, eps, mode):
        x = x.contiguous()
        y, mean, rstd = K.space_depth_ln_forward(x, mode, w, b, eps)
        ctx.mode = mode
        ctx.save_for_backward(x, mean, rstd, w)
        ctx.direct = DEFERRED.direct(w, b)
        if ctx.direct:
            ctx.params, ctx.gen = (w, b), DEFERRED.note_params(w, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, w = ctx.saved_tensors
        if ctx.direct:
            dx, _, _ = K.space_depth_ln_backward(dy, x, ctx.mode, mean, rstd, w, into=tuple(DEFERRED.grad_of(p) for p in ctx.params))
            DEFERRED.written(ctx.gen, *ctx.params)
            return dx, None, None, None, None
        dx, dw, db = K.space_depth_ln_backward(dy, x, ctx.mode, mean, rstd, w)
        return dx, dw, db, None, None


def space_depth_layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float, mode: str) -> torch.Tensor:
    """x (B,T,H,W,C) -> LayerNorm rows of the 'down' (B,T,H/2,W/2,4C) or 'up' (B,T,2H,2W,C/4) rearrangement, order (neiw neih c)."""
    B, T, H, W, C = x.shape
    y = _SpaceDepthLayerNorm.apply(x.reshape(B * T, H, W, C), w, b, eps, mode)
    return y.reshape(B, T, *y.shape[1:])


class _LayerNormSkip(_Fn):
    """(LayerNorm(x), x): the second output is x itself, for the skip connection around the normalised branch.  Both gradients meet in
    THIS node's backward, where the LayerNorm backward kernel adds the skip gradient on its way out (vmg_layernorm_bwd_add) -- autograd
    would otherwise sum the two with a separate full-size pass per LayerNorm."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        x = x.contiguous()
        y, mean, rstd = K.layernorm_forward(x, w, b, eps)
        ctx.save_for_backward(x, mean, rstd, w)
        ctx.direct = DEFERRED.direct(w, b)
        if ctx.direct:
            ctx.params, ctx.gen = (w, b), DEFERRED.note_params(w, b)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        x, mean, rstd, w = ctx.saved_tensors
        into = tuple(DEFERRED.grad_of(p) for p in ctx.params) if ctx.direct else None
        if dy is None:  # (the normalised branch is unused)
            return dskip, None, None, None
        dx, dw, db = K.layernorm_backward(dy, x, mean, rstd, w, into=into, add=dskip)
        if ctx.direct:
            DEFERRED.written(ctx.gen, *ctx.params)
            return dx, None, None, None
        return dx, dw, db, None


def layer_norm_skip(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5):
    """(nn.LayerNorm(x), x) with the skip gradient summed inside the LayerNorm backward kernel."""
    return _LayerNormSkip.apply(x, w, b, eps)


class _LayerNormFan(_Fn):
    """(y_1 .. y_n, x): n handles of the SAME LayerNorm(x) for n consumers, and x for the skip connection.  Autograd hands this node's backward
    all n output gradients at once; the LayerNorm backward kernel sums them (and the skip gradient) while it reads them
    (vmg_layernorm_bwd_multi) -- the n - 1 full-size adds autograd would issue for a tensor with n consumers never run."""

    @staticmethod
    def forward(ctx, x, w, b, eps, n):
        x = x.contiguous()
        y, mean, rstd = K.layernorm_forward(x, w, b, eps)
        ctx.save_for_backward(x, mean, rstd, w)
        ctx.direct = DEFERRED.direct(w, b)
        if ctx.direct:
            ctx.params, ctx.gen = (w, b), DEFERRED.note_params(w, b)
        ctx.set_materialize_grads(False)
        return (*[y.view_as(y) for _ in range(n)], x.view_as(x))

    @staticmethod
    def backward(ctx, *grads):
        x, mean, rstd, w = ctx.saved_tensors
        dys = [g for g in grads[:-1] if g is not None]
        dskip = grads[-1]
        if not dys:  # (the normalised branch is unused)
            if ctx.direct:
                DEFERRED.written(ctx.gen, *ctx.params)
            return dskip, None, None, None, None
        into = tuple(DEFERRED.grad_of(p) for p in ctx.params) if ctx.direct else None
        dx, dw, db = K.layernorm_backward(dys, x, mean, rstd, w, into=into, add=dskip)
        if ctx.direct:
            DEFERRED.written(ctx.gen, *ctx.params)
            return dx, None, None, None, None
        return dx, dw, db, None, None


def layer_norm_fan(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float, n: int):
    """([LayerNorm(x)] * n, x): see _LayerNormFan (n <= 5)."""
    out = _LayerNormFan.apply(x, w, b, eps, int(n))
    return list(out[:-1]), out[-1]


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """nn.LayerNorm over the last (channel) dimension."""
    return _LayerNorm.apply(x, w, b, eps)


# =========================================================================================================
# Ops still expressed with PyTorch-ROCm device ops (autograd by torch).  Each is HBM-bound index / elementwise
# work around the MFMA kernels above and is replaced by a hand-written HIP kernel as the rounds proceed; the
# table in DESIGN.md ("what runs where") is kept in step with this section.  None of them runs on the CPU.
# =========================================================================================================
import math

import torch.nn.functional as F


class _SplitHalves(_Fn):
    """x (2n, ...) -> (x[:n], x[n:]) as fresh tensors; the backward writes both gradients into one tensor with elementwise kernels.  Plain
    slicing has a SliceBackward that copies each gradient into a zero tensor with a device-to-device memcpy, which a captured step cannot
    hand to csrc/replay.hip (used on the small flow fields of VMG.compute_flow)."""

    @staticmethod
    def forward(ctx, x):
        n = x.shape[0] // 2
        return torch.mul(x[:n], 1), torch.mul(x[n:], 1)

    @staticmethod
    def backward(ctx, ga, gb):
        n = ga.shape[0]
        out = ga.new_empty((2 * n,) + tuple(ga.shape[1:]))
        torch.mul(ga, 1, out=out[:n])
        torch.mul(gb, 1, out=out[n:])
        return out


def split_halves(x: torch.Tensor):
    return _SplitHalves.apply(x)


_PAIR_INDEX = {}


def _pair_index(n: int, t: int, device):
    """index tables of pair_frames: forward (t * 2n, 1) into the (n, t) frames, backward (n * t, 2) into the (t, 2n) frames."""
    key = (n, t, str(device))
    tabs = _PAIR_INDEX.get(key)
    if tabs is None:
        fwd = [[r * t + (t - 1 - j)] if r < n else [(r - n) * t + j] for j in range(t) for r in range(2 * n)]
        bwd = [[(t - 1 - i) * 2 * n + b, i * 2 * n + n + b] for b in range(n) for i in range(t)]
        tabs = _PAIR_INDEX[key] = (torch.tensor(fwd, dtype=torch.int32, device=device), torch.tensor(bwd, dtype=torch.int32, device=device))
    return tabs


class _PairFrames(_Fn):
    @staticmethod
    def forward(ctx, x):
        n, t = x.shape[:2]
        ctx.n, ctx.t, ctx.frame = n, t, tuple(x.shape[2:])
        fi, _ = _pair_index(n, t, x.device)
        return K.frame_gather(x.contiguous(), fi, 2 * n * t, ctx.frame).view(t, 2 * n, *ctx.frame)

    @staticmethod
    def backward(ctx, g):
        n, t = ctx.n, ctx.t
        _, bi = _pair_index(n, t, g.device)
        return K.frame_gather(g.contiguous(), bi, n * t, ctx.frame).view(n, t, *ctx.frame)


def pair_frames(x: torch.Tensor) -> torch.Tensor:
    """x (n, t, ...) batch-major -> (t, 2n, ...): row j = [frame t-1-j of every clip | frame j of every clip] -- what step j of the two
    direction sweeps of the recurrence works on (model.Trajectory_multi_head).  One gather; the gradient is one gather-add
    (dx[b, i] = g[t-1-i, b] + g[i, n+b], fp32 sum).  torch spells it transpose + flip + cat: three passes each way."""
    return _PairFrames.apply(x)


class _PairFrameSteps(_Fn):
    @staticmethod
    def forward(ctx, x):
        n, t = x.shape[:2]
        ctx.n, ctx.t, ctx.shape = n, t, tuple(x.shape)
        fi, _ = _pair_index(n, t, x.device)
        return tuple(K.frame_gather(x.contiguous(), fi, 2 * n * t, tuple(x.shape[2:])).view(t, 2 * n, *x.shape[2:]).unbind(0))

    @staticmethod
    def backward(ctx, *gs):
        n, t = ctx.n, ctx.t
        ref = next(g for g in gs if g is not None)
        gs = [torch.zeros_like(ref) if g is None else g.contiguous() for g in gs]
        dx = torch.empty(ctx.shape, dtype=ref.dtype, device=ref.device)
        K.pair_steps(2, gs, dx, None, n, t)
        return dx


def pair_frame_steps(x: torch.Tensor):
    """pair_frames as a tuple of its t step tensors: the step gradients (separate allocations, one per recurrence step) are summed straight into dx by
    one kernel -- `pair_frames(x).unbind(0)` made autograd stack them into one (t, 2n, ...) tensor first (66 MB at the bench shape)."""
    return _PairFrameSteps.apply(x)


class _UnpairSteps(_Fn):
    @staticmethod
    def forward(ctx, n, *steps):
        t = len(steps)
        steps = [s.contiguous() for s in steps]
        frame = tuple(steps[0].shape[1:])
        back = torch.empty((n, t, *frame), dtype=steps[0].dtype, device=steps[0].device)
        fwd = torch.empty_like(back)
        K.pair_steps(0, steps, back, fwd, n, t)
        ctx.n, ctx.t, ctx.frame = n, t, frame
        return back, fwd

    @staticmethod
    def backward(ctx, dback, dfwd):
        n, t = ctx.n, ctx.t
        ref = dback if dback is not None else dfwd
        dback = torch.zeros((n, t, *ctx.frame), dtype=ref.dtype, device=ref.device) if dback is None else dback.contiguous()
        dfwd = torch.zeros_like(dback) if dfwd is None else dfwd.contiguous()
        buf = torch.empty((t, 2 * n, *ctx.frame), dtype=ref.dtype, device=ref.device)
        gs = buf.unbind(0)
        K.pair_steps(1, gs, dback, dfwd, n, t)
        return (None, *gs)


def unpair_steps(steps: Sequence[torch.Tensor], n: int):
    """The t step outputs of the lock-step recurrence ((2n, ...) each: rows [0, n) the backward sweep at frame t-1-j, rows [n, 2n) the forward sweep at
    frame j) -> (back, fwd), both (n, t, ...) in frame order (models/trajectory.py:394-395 `insert(0, ...)`, :479 `append`, then the stacks): one kernel
    each way instead of t splits + two stacks forward and t concatenations backward."""
    return _UnpairSteps.apply(n, *steps)


def morph_tokens(x: torch.Tensor, axis: str, chunk: int, Cp: int) -> torch.Tensor:
    """Token layout of the H-/W-branch (models/function.py:763-764, 776-777): pad C->Cp and the mixed axis to a
    multiple of `chunk`; token (group, k) gets features f = p*S + s <- x[position p of the group, channel k*S + s]."""
    B, T, H, W, C = x.shape
    S = Cp // chunk
    if axis == "h":
        Hp = int(math.ceil(H / chunk)) * chunk
        xp = (F.pad(x, (0, Cp - C, 0, 0, 0, Hp - H)) if (Cp != C or Hp != H) else x).transpose(2, 3)  # (F.pad by nothing is a full copy)
        L = W * Hp
    else:
        Wp = int(math.ceil(W / chunk)) * chunk
        xp = F.pad(x, (0, Cp - C, 0, Wp - W)) if (Cp != C or Wp != W) else x
        L = H * Wp
    t = xp.reshape(B, T, L // chunk, chunk, chunk, S)
    return t.permute(0, 1, 2, 4, 3, 5).reshape(B, T, L // chunk, chunk, chunk * S).contiguous()


def morph_untokens(t: torch.Tensor, axis: str, chunk: int, Cp: int, H: int, W: int, C: int) -> torch.Tensor:
    """Inverse layout + crop (models/function.py:772, 785)."""
    B, T, G = t.shape[:3]
    S = Cp // chunk
    u = t.reshape(B, T, G, chunk, chunk, S).permute(0, 1, 2, 4, 3, 5)
    if axis == "h":
        Hp = int(math.ceil(H / chunk)) * chunk
        return u.reshape(B, T, W, Hp, Cp).transpose(2, 3)[..., 0:H, :, :C].contiguous()
    Wp = int(math.ceil(W / chunk)) * chunk
    return u.reshape(B, T, H, Wp, Cp)[..., 0:W, :C].contiguous()


class _MorphLinear(_Fn):
    """One MorphFC branch: relu(tokens(x) W^T + b) / Cp back in pixel layout (models/function.py:763-772, 776-785) as ONE kernel with the
    token reshuffle in the GEMM's addressing (vmg_morphfc_fwd).  Backward: the data gradient is the same kernel on (dy * relu'(y) / Cp)
    with the transposed weight; the weight gradient's two token matrices are side outputs of those two launches (the lanes write the
    MFMA fragments they hold), summed by the batched Linear weight-gradient GEMM."""

    @staticmethod
    def forward(ctx, x, weight, bias, axis, chunk, Cp):
        x = x.contiguous()
        nct = (Cp + 15) // 16
        need_w = ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2])
        # the kernel writes the token matrix it multiplies as a side output: the weight gradient's x operand without a gather pass
        ctx.tok_mode = bool(need_w and ctx.needs_input_grad[0] and Cp % 8 == 0)
        y, tok = K.morphfc_forward(x, axis, chunk, Cp, packed(weight, x.dtype, "fwd", [Cp], tiles=nct), bias, True, 1.0, 1.0 / Cp, want_tokens=ctx.tok_mode)
        ctx.cfg = (axis, chunk, Cp)
        ctx.has_bias = bias is not None
        ctx.defer = DEFERRED.mode == "deferred" and isinstance(weight, torch.nn.Parameter) and ctx.needs_input_grad[1] and \
            (bias is None or isinstance(bias, torch.nn.Parameter))
        if ctx.defer:
            ctx.gen = DEFERRED.note_use(weight, bias)
            ctx.bias_ref = bias
        ctx.save_for_backward(tok if ctx.tok_mode else x, y, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight = ctx.saved_tensors  # (x: the token matrix in tok_mode)
        axis, chunk, Cp = ctx.cfg
        dy = dy.contiguous()
        nct = (Cp + 15) // 16
        dx = dpre = None
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        if ctx.needs_input_grad[0]:
            dx, dpre = K.morphfc_forward(dy, axis, chunk, Cp, packed(weight, dy.dtype, "dgrad", None, 0, Cp, tiles=nct), None, False, 1.0 / Cp, 1.0, mask=y,
                                         want_tokens=ctx.tok_mode and need_w)
        d_w = d_b = None
        if need_w:
            if ctx.tok_mode:
                tok = x
            else:
                tok = morph_tokens(x, axis, chunk, Cp)                                                    # (B,T,G,chunk,Cp)
                dpre = morph_tokens(K.act_backward(dy, y, hip.ACT_RELU, 0.0, 1.0 / Cp), axis, chunk, Cp)
            M = tok.numel() // Cp
            if ctx.defer:
                DEFERRED.add(weight, ctx.bias_ref, [tok], [Cp], dpre, 1, 1, 1, M, gen=ctx.gen)
            else:
                d_w, d_b = _wgrad_now(weight, ctx.has_bias and ctx.needs_input_grad[2], [tok], [Cp], dpre, 1, 1, 1, M)
        return dx, d_w, d_b, None, None, None


class _MorphGather(_Fn):
    """x (B,T,H,W,C) -> token matrix (rows, ld) by ONE gather kernel (the reference's pad + rearrange chain, models/function.py:749-750,
    763-764, 776-777); ld = Cp rounded up to 8, features [Cp, ld) are zeros (the convolution kernel reads 16-byte vectors).  Backward = the
    scatter kernel (the two are each other's adjoint)."""

    @staticmethod
    def forward(ctx, x, axis, chunk, Cp):
        x = x.contiguous()
        ctx.cfg = (axis, chunk, Cp, tuple(x.shape))
        return K.morph_tokens_gather(x, axis, chunk, Cp, _pad_to(Cp))

    @staticmethod
    def backward(ctx, g):
        axis, chunk, Cp, shape = ctx.cfg
        return K.morph_tokens_scatter(g.contiguous(), axis, chunk, Cp, shape), None, None, None


class _MorphScatter(_Fn):
    """token matrix (rows, Cp) -> (B,T,H,W,C) by ONE scatter kernel (inverse layout + crop, models/function.py:772, 785); backward = the gather."""

    @staticmethod
    def forward(ctx, tok, axis, chunk, Cp, shape):
        ctx.cfg = (axis, chunk, Cp, tuple(shape))
        return K.morph_tokens_scatter(tok.contiguous(), axis, chunk, Cp, shape)

    @staticmethod
    def backward(ctx, g):
        axis, chunk, Cp, shape = ctx.cfg
        return K.morph_tokens_gather(g.contiguous(), axis, chunk, Cp, Cp), None, None, None, None


def morph_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], axis: str, chunk: int, Cp: int) -> torch.Tensor:
    """relu(Linear(tokens)) / Cp of one MorphFC branch on (B,T,H,W,C).  Fused kernel where it is instantiated (stage 0 of both shipped
    configurations); the general path -- any chunk / Cp, fp32 too -- is gather kernel + Linear kernel + scatter kernel."""
    if K.morph_fused_ok(x, chunk, Cp):
        return _MorphLinear.apply(x, weight, bias, axis, chunk, Cp)
    tok = _MorphGather.apply(x, axis, chunk, Cp)
    if tok.shape[-1] != Cp:  # (Cp = 228: the Linear takes the Cp real features; its kernel reads the zero-padded matrix they are a slice of)
        padded, tok = tok, tok[:, :Cp]
        tok._vmg_padded = padded
    t = linear(tok, weight, bias, act=hip.ACT_RELU, alpha=1.0 / Cp)
    return _MorphScatter.apply(t, axis, chunk, Cp, tuple(x.shape))


class _ChannelAttention(_Fn):
    """(r * sigmoid(W2 relu(W1 GAP(r) + b1) + b2) + x) * s on (N,H,W,C): CALayer + RCAB residual (models/function.py:555-558,
    581).  The two full-tensor passes (GAP reduction, scale+residual) are HIP kernels; the (N,C)-sized squeeze-excite MLP
    and its backward are a handful of tiny fp32 ops."""

    @staticmethod
    def forward(ctx, r, x, w1, b1, w2, b2, s):
        r, x = r.contiguous(), x.contiguous()
        N, C = r.shape[0], r.shape[-1]
        R = r.numel() // (N * C)
        m = K.group_reduce(r, N, scale=1.0 / R)
        pre, g = K.se_mlp_forward(m, w1.flatten(1), b1, w2.flatten(1), b2, hip.ACT_RELU, 0)
        out = K.tab_elementwise(K.OP_CA_FWD, r, x, coef=g, s=s, G=N)
        ctx.s, ctx.R = s, R
        ctx.save_for_backward(r, g, m, pre, w1, w2)
        ctx.direct = DEFERRED.direct(w1, b1, w2, b2)
        if ctx.direct:
            ctx.params, ctx.gen = (w1, b1, w2, b2), DEFERRED.note_params(w1, b1, w2, b2)
        return out

    @staticmethod
    def backward(ctx, dy):
        r, g, m, pre, w1, w2 = ctx.saved_tensors
        s, R = ctx.s, ctx.R
        dy = dy.contiguous()
        N = r.shape[0]
        dg = K.group_reduce(dy, N, b=r, mode=1, scale=s)          # d out / d g summed over pixels
        into = tuple(DEFERRED.grad_of(p) for p in ctx.params) if ctx.direct else None
        dm, dw1, db1, dw2, db2 = K.se_mlp_backward(dg, g, m, pre, w1.flatten(1), w2.flatten(1), hip.ACT_RELU, 0, 1.0 / R, into=into)  # dm: gradient of the GAP output / R
        d_r, d_x = K.tab_elementwise(K.OP_CA_BWD, dy, coef=g, add=dm, s=s, G=N, nout=2)
        if ctx.direct:
            DEFERRED.written(ctx.gen, *ctx.params)
            return d_r, d_x, None, None, None, None, None
        return (d_r, d_x, dw1.reshape(w1.shape), db1, dw2.reshape(w2.shape), db2, None)


def channel_attention_residual(r, x, w1, b1, w2, b2, out_scale: float):
    return _ChannelAttention.apply(r, x, w1, b1, w2, b2, float(out_scale))

class _ResidualDropPath(_Fn):
    """res + y * g with g a per-(sample, channel) fp32 coefficient: the TAB residuals `x + DropPath(y) * s`
    (models/function.py:1212-1217) in ONE pass (HIP, the channel-attention scale kernel) instead of mask-multiply, scale and
    add; the backward is one multiply (d_res is dy itself)."""

    @staticmethod
    def forward(ctx, res, y, g, gb=None):
        out = K.tab_elementwise(K.OP_CA_FWD, y.contiguous(), res.contiguous(), coef=g, s=1.0, G=g.shape[0])
        ctx.save_for_backward(g)
        return out

    @staticmethod
    def backward(ctx, dy):
        (g,) = ctx.saved_tensors  # (B, C) fp32; the product is rounded once, like dy * g.to(dtype) (g takes the values 0 and s / keep)
        dy = dy.contiguous()
        return dy, K.tab_elementwise(K.OP_SCALE, dy, coef=g.contiguous(), s=1.0, G=g.shape[0]), None, None


class _DropPlan:
    """The DropPath coefficients of a whole forward pass from one handful of launches.  Every residual_drop_path call used to cost five
    tiny kernels (bernoulli_, div_, mul, expand + contiguous, a cast for the backward) -- ~100 launches per train step.  The calls of
    one forward pass (batch, channels, keep probability, scale) are recorded; the NEXT pass, if it makes the same calls in the same
    order, draws all masks at once (rand < keep, / keep, * scale: timm's DropPath per call, from one generator draw).  Any
    deviation from the recorded sequence falls back to the per-call path."""

    def __init__(self):
        self.rec, self.plan, self.g, self.gb, self.idx, self.consts, self.offs = [], None, None, None, 0, {}, []

    def begin(self, device, active: bool):
        rec, self.rec = self.rec, []
        self.idx, self.g, self.gb, self.plan = 0, None, None, None
        if not active or not rec or any(r[2] <= 0.0 for r in rec):
            return
        key = (tuple(rec), str(device))
        c = self.consts.get(key)
        if c is None:
            # one row of constants per (call, sample); `rows` expands a call's per-sample coefficients over its channels -- the calls of a pass
            # may differ in batch and channel count (the stages of the full configuration: 112 / 224 / 448 channels), so the coefficients
            # live in ONE flat buffer, call i at offs[i] as (B_i, C_i)
            keep = torch.tensor([r[2] for r in rec for _ in range(r[0])], dtype=torch.float32, device=device)
            mult = torch.tensor([r[3] / r[2] for r in rec for _ in range(r[0])], dtype=torch.float32, device=device)
            rows, offs, row0, off = [], [], 0, 0
            for B, C, _, _ in rec:
                rows.append((torch.arange(B, dtype=torch.int64) + row0).repeat_interleave(C))
                offs.append(off)
                row0 += B
                off += B * C
            c = self.consts[key] = (keep, mult, torch.cat(rows).to(device), offs)
            if len(self.consts) > 8:
                self.consts.pop(next(iter(self.consts)))
        u = torch.rand(c[0].numel(), device=device)
        self.g = ((u < c[0]).to(torch.float32) * c[1]).index_select(0, c[2])
        self.offs = c[3]
        self.plan = rec

    def take(self, B, C, keep, scale, dtype):
        i = self.idx
        self.idx += 1
        self.rec.append((B, C, float(keep), float(scale)))
        if self.g is None or i >= len(self.plan) or self.plan[i] != self.rec[-1]:
            self.g = None  # out of step with the recorded sequence: per-call path for the rest of this pass
            return None, None
        if self.gb is None or self.gb.dtype != dtype:
            self.gb = self.g.to(dtype)
        o = self.offs[i]
        return self.g[o:o + B * C].view(B, C), self.gb[o:o + B * C].view(B, C)


DROP = _DropPlan()


def residual_drop_path(res: torch.Tensor, y: torch.Tensor, p: float, training: bool, scale: float = 1.0) -> torch.Tensor:
    """res + DropPath_p(y) * scale with timm's DropPath semantics (per-sample Bernoulli(1-p) mask / (1-p), sample = dim 0)."""
    if (p == 0.0 or not training) and scale == 1.0:
        return res + y
    B, C = y.shape[0], y.shape[-1]
    if p > 0.0 and training:
        keep = 1.0 - p
        g, gb = DROP.take(B, C, keep, scale, y.dtype)
        if g is not None:
            return _ResidualDropPath.apply(res, y, g, gb)
        mask = torch.empty(B, 1, dtype=torch.float32, device=y.device).bernoulli_(keep)
        if keep > 0.0:
            mask.div_(keep)
        g = (mask * scale).expand(B, C).contiguous()
    else:
        g = torch.full((B, C), float(scale), dtype=torch.float32, device=y.device)
    return _ResidualDropPath.apply(res, y, g)



class _ReweightMix(_Fn):
    """Softmax re-weighting of the three mixer branches (models/function.py:791-793):
    a = softmax_3(Mlp(mean_{T,H,W}(h + w + c)));  y = h*a0 + w*a1 + c*a2."""

    @staticmethod
    def forward(ctx, h, w, c, fc1w, fc1b, fc2w, fc2b):
        h, w, c = h.contiguous(), w.contiguous(), c.contiguous()
        B, C = h.shape[0], h.shape[-1]
        R = h.numel() // (B * C)
        m = K.group_reduce(h, B, b=w, c3=c, scale=1.0 / R)
        pre, a = K.se_mlp_forward(m, fc1w, fc1b, fc2w, fc2b, hip.ACT_GELU, 1)  # a (B, 3C) = (B, C, 3): softmax over each channel's three logits
        y = K.tab_elementwise(K.OP_MIX_FWD, h, w, c, coef=a, G=B)
        ctx.R = R
        ctx.save_for_backward(h, w, c, a, m, pre, fc1w, fc2w)
        ctx.direct = DEFERRED.direct(fc1w, fc1b, fc2w, fc2b)
        if ctx.direct:
            ctx.params, ctx.gen = (fc1w, fc1b, fc2w, fc2b), DEFERRED.note_params(fc1w, fc1b, fc2w, fc2b)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, w, c, a, m, pre, fc1w, fc2w = ctx.saved_tensors
        R = ctx.R
        dy = dy.contiguous()
        B, C = h.shape[0], h.shape[-1]
        da = K.group_reduce3(dy, h, w, c, B).reshape(B, 3 * C)  # (B,C,3): d loss / d softmax weights, one pass over dy
        into = tuple(DEFERRED.grad_of(p) for p in ctx.params) if ctx.direct else None
        dm, dw1, db1, dw2, db2 = K.se_mlp_backward(da, a, m, pre, fc1w, fc2w, hip.ACT_GELU, 1, 1.0 / R, into=into)   # softmax, Linear, GELU, Linear backward
        dh, dw, dc = K.tab_elementwise(K.OP_MIX_BWD, dy, coef=a, add=dm, G=B, nout=3)
        if ctx.direct:
            DEFERRED.written(ctx.gen, *ctx.params)
            return dh, dw, dc, None, None, None, None
        return dh, dw, dc, dw1, db1, dw2, db2


def reweight_mix(h, w, c, fc1w, fc1b, fc2w, fc2b):
    return _ReweightMix.apply(h, w, c, fc1w, fc1b, fc2w, fc2b)


class _TanhGate(_Fn):
    """(x + y) * tanh(y) (models/function.py:801-802)."""

    @staticmethod
    def forward(ctx, x, y):
        x, y = x.contiguous(), y.contiguous()
        ctx.save_for_backward(x, y)
        return K.tab_elementwise(K.OP_GATE_FWD, x, y)

    @staticmethod
    def backward(ctx, d):
        x, y = ctx.saved_tensors
        dx, dy = K.tab_elementwise(K.OP_GATE_BWD, d.contiguous(), x, y, nout=2)
        return dx, dy


def tanh_gate(x, y):
    return _TanhGate.apply(x, y)


class _GateResidual(_Fn):
    """res + DropPath((x + y) * tanh(y)) * s: the mixer's symmetric gate (models/function.py:801-802) and the TAB residual around the mixer
    (function.py:1212-1214) in ONE pass each way -- the gated tensor is neither written nor read back (round 3 ran tanh_gate, then
    residual_drop_path: two passes forward, two backward, per TAB).  Same bits as the two-pass form: the gate is rounded to the tensor dtype
    before the residual coefficient is applied, as when it was stored."""

    @staticmethod
    def forward(ctx, x, y, res, g):
        x, y, res = x.contiguous(), y.contiguous(), res.contiguous()
        ctx.save_for_backward(x, y, g)
        return K.tab_elementwise(K.OP_GATE_RES_FWD, x, y, res, coef=g, s=1.0, G=g.shape[0])

    @staticmethod
    def backward(ctx, d):
        x, y, g = ctx.saved_tensors
        d = d.contiguous()
        dx, dy = K.tab_elementwise(K.OP_GATE_RES_BWD, d, x, y, coef=g, s=1.0, G=g.shape[0], nout=2)
        return dx, dy, d, None


_ONES = {}


def gate_residual(x: torch.Tensor, y: torch.Tensor, res: torch.Tensor, p: float, training: bool, scale: float = 1.0) -> torch.Tensor:
    """res + DropPath_p((x + y) * tanh(y)) * scale (timm DropPath semantics, see residual_drop_path) as one kernel."""
    B, C = y.shape[0], y.shape[-1]
    if p > 0.0 and training:
        keep = 1.0 - p
        g, _ = DROP.take(B, C, keep, scale, y.dtype)
        if g is None:
            mask = torch.empty(B, 1, dtype=torch.float32, device=y.device).bernoulli_(keep)
            if keep > 0.0:
                mask.div_(keep)
            g = (mask * scale).expand(B, C).contiguous()
    else:
        key = (B, C, float(scale), str(y.device))
        g = _ONES.get(key)
        if g is None:
            g = _ONES[key] = torch.full((B, C), float(scale), dtype=torch.float32, device=y.device)
    return _GateResidual.apply(x, y, res, g)


class _MaxPool(_Fn):
    @staticmethod
    def forward(ctx, x, f):
        y, idx = K.maxpool_forward(x.contiguous(), f)
        ctx.f = f
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return K.maxpool_backward(dy, idx, ctx.f), None


def max_pool(x: torch.Tensor, f: int) -> torch.Tensor:
    """adaptive_max_pool2d(x, (H/f, W/f)) on channels-last (n,h,w,c) when f divides h and w (models/vmg.py:519, 525)."""
    return _MaxPool.apply(x, int(f))


class _GroupNorm1ReLU(_Fn):
    """relu(GroupNorm(1, C)(x)) on channels-last (n,h,w,c): per-sample statistics over (h,w,c), per-channel affine
    (models/vmg.py:390-399).  Two grouped reductions + one coefficient-broadcast elementwise pass each way; the (n, c)-sized
    algebra in between is a handful of tiny fp32 ops."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        x = x.contiguous()
        n, c = x.shape[0], x.shape[-1]
        cnt = x.numel() // n
        s1 = K.group_reduce(x, n).sum(1)
        s2 = K.group_reduce(x, n, b=x, mode=1).sum(1)
        mu = s1 / cnt
        rs = torch.rsqrt((s2 / cnt - mu * mu).clamp_min(0) + eps)
        coef = torch.stack([rs[:, None] * w[None], torch.zeros(n, c, device=x.device)], -1).contiguous()
        add = (b[None] - mu[:, None] * rs[:, None] * w[None]).contiguous()
        y = K.tab_elementwise(K.OP_AFFINE2, x, coef=coef, add=add, s=1.0, G=n)
        ctx.save_for_backward(x, y, w, mu, rs)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, w, mu, rs = ctx.saved_tensors
        n, c = x.shape[0], x.shape[-1]
        cnt = x.numel() // n
        g = K.act_backward(dy.contiguous(), y, hip.ACT_RELU, 0.0, 1.0)   # dy * relu'(y)
        S1 = K.group_reduce(g, n)                                         # (n, c): sum_r g
        S2 = K.group_reduce(g, n, b=x, mode=1)                            # (n, c): sum_r g * x
        gxh = rs[:, None] * (S2 - mu[:, None] * S1)                       # sum_r g * xhat
        dw, db = gxh.sum(0), S1.sum(0)
        m1 = (S1 * w[None]).sum(1) / cnt                                  # mean(g * w)
        m2 = (gxh * w[None]).sum(1) / cnt                                 # mean(g * w * xhat)
        coef = torch.stack([(rs[:, None] * w[None]).expand(n, c), (-rs * rs * m2)[:, None].expand(n, c)], -1).contiguous()
        add = (rs * (mu * rs * m2 - m1))[:, None].expand(n, c).contiguous()
        dx = K.tab_elementwise(K.OP_AFFINE2, g, x, coef=coef, add=add, s=0.0, G=n)
        return dx, dw, db, None


def group_norm1_relu(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    return _GroupNorm1ReLU.apply(x, w, b, float(eps))


class _Upsample2xAC(_Fn):
    """scale * F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True) on channels-last fp32 (SPyNet's flow between
    pyramid levels, models/vmg.py:97-102)."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return K.upsample2x_ac(x.contiguous(), scale)

    @staticmethod
    def backward(ctx, dy):
        return K.upsample2x_ac(dy.contiguous(), ctx.scale, backward=True), None


def upsample2x_flow(x: torch.Tensor, scale: float = 2.0) -> torch.Tensor:
    return _Upsample2xAC.apply(x, float(scale))


class _SpyOperand(_Fn):
    @staticmethod
    def forward(ctx, ref, warped, up):
        return K.spy_operand(ref.contiguous(), warped.contiguous(), up.contiguous())

    @staticmethod
    def backward(ctx, d):
        dwarped, dup = K.spy_operand_backward(d.contiguous())
        return None, dwarped, dup


def spy_operand(ref: torch.Tensor, warped: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """One pyramid level's network input `cat([ref, warp(supp, flow_up), flow_up])` (models/vmg.py:76-84) as ONE 8-channel tensor [ref RGB | warped RGB |
    flow] from the 8-channel images (RGB in front) and the fp32 flow; the reference image takes no gradient (the pyramid is built under no_grad)."""
    return _SpyOperand.apply(ref, warped, up)


class _SpyFlowAdd(_Fn):
    @staticmethod
    def forward(ctx, up, res):
        ctx.dt = res.dtype
        return K.spy_flow_add(up.contiguous(), res.contiguous())

    @staticmethod
    def backward(ctx, g):
        return g, g.to(ctx.dt)


def spy_flow_add(up: torch.Tensor, res: torch.Tensor) -> torch.Tensor:
    """flow = flow_up + basic_module(...) (models/vmg.py:85): fp32 sum of the fp32 flow and the residual in the compute dtype, one pass."""
    return _SpyFlowAdd.apply(up, res)


def identity_grid(n: int, h: int, w: int, device) -> torch.Tensor:
    ys, xs = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
    return torch.stack([xs, ys], 0).float()[None].expand(n, -1, -1, -1).contiguous()


def _norm_grid(g, h, w):
    gx = 2.0 * g[..., 0] / max(w - 1, 1) - 1.0
    gy = 2.0 * g[..., 1] / max(h - 1, 1) - 1.0
    return torch.stack((gx, gy), -1)


class _WarpBilinear(_Fn):
    @staticmethod
    def forward(ctx, x, flow):
        x = x.contiguous()
        flow = flow.contiguous()
        ctx.save_for_backward(x, flow)
        return K.warp_bilinear_forward(x, flow)

    @staticmethod
    def backward(ctx, dy):
        x, flow = ctx.saved_tensors
        return K.warp_bilinear_backward(x, flow, dy)


def grid_sample_flow(x: torch.Tensor, flow: torch.Tensor, mode: str, padding: str) -> torch.Tensor:
    """Channels-last flow warp: x (n,h,w,c), flow (n,h,w,2) fp32 pixel offsets (models/trajectory.py:95-116).
    The recurrence only uses bilinear + border (trajectory.py:330, 414)."""
    if mode != "bilinear" or padding != "border":
        raise HipError("the HIP flow warp implements bilinear / border (the only combination on the path)")
    return _WarpBilinear.apply(x, flow.float())


class _FlowSmooth(_Fn):
    @staticmethod
    def forward(ctx, flow, r):
        ctx.r = r
        return K.flow_smooth(flow.contiguous(), r)

    @staticmethod
    def backward(ctx, g):
        return K.flow_smooth(g.contiguous(), ctx.r, backward=True), None


def flow_smooth(flow: torch.Tensor, r: int) -> torch.Tensor:
    """Mlp_encoder's flow smoothing (models/function.py:1466-1478): reflect-pad to a multiple of r, r x r mean, nearest x r, crop -- one kernel each way on the
    (..., H, W) fp32 flow planes."""
    return _FlowSmooth.apply(flow.float(), int(r))


def warp_locations(loc: torch.Tensor, flow: torch.Tensor) -> torch.Tensor:
    """Advect the tracked-location maps (n,2k,h,w) with nearest sampling, border padding (trajectory.py:332-333).
    Not differentiable (the reference's nearest sampling has zero gradient w.r.t. grid and the maps are constants)."""
    return K.warp_nearest_planes(loc.detach(), flow.detach().contiguous())


class _FanOut(_Fn):
    @staticmethod
    def forward(ctx, x, k):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(k))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        g0 = gs[0]
        vn = 8 if g0.dtype == torch.bfloat16 else 4
        if len(gs) <= 4 and g0.dtype in (torch.bfloat16, torch.float32) and g0.numel() % vn == 0 and all(g.dtype == g0.dtype and g.shape == g0.shape for g in gs):
            return K.sum_n([g.contiguous() for g in gs]), None
        tot = gs[0]
        for g in gs[1:]:
            tot = tot + g
        return tot, None


def fan_out(x: torch.Tensor, k: int):
    """k handles on x, one per consumer: the gradient of x is then ONE sum over the consumers' gradients (fp32, one rounding, vmg_sum_n) instead of autograd's
    k - 1 pairwise adds.  Without gradients: x itself, k times."""
    if k == 1 or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * k
    return _FanOut.apply(x, k)


class _GradBank:
    """Gradient accumulator of ONE key / value frame over all the trajectory-attention calls that attend to it (a key-frame of a 7-frame clip
    is attended by up to 6 later frames): their backward kernels scatter into the same FP32 buffer with float atomics, instead of each call
    zero-filling buffers of its own that autograd then sums pairwise; the sum is rounded to the frame's dtype once, when autograd reaches the
    frame (_Banked.backward)."""
    __slots__ = ("buf", "dtype")

    def __init__(self, dtype):
        self.buf = None
        self.dtype = dtype


class _Banked(_Fn):
    """Identity.  Backward: what the attention calls scattered into the bank (+ the gradient of any other use of the output)."""

    @staticmethod
    def forward(ctx, x, bank):
        ctx.bank = bank
        ctx.set_materialize_grads(False)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        buf, ctx.bank.buf = ctx.bank.buf, None
        if buf is None:
            return g, None
        dt = ctx.bank.dtype
        if dt != torch.float32 and buf.numel() % 4 == 0 and (g is None or g.dtype == dt):
            # ONE rounding of the finished sum (+ the gradient of the frame's other uses, added in fp32), and the accumulator goes back to the pool cleared
            out = K.cast_clear(buf, dt, add=g.contiguous() if g is not None else None)
            K.ACC_POOL.give(buf)
            return out, None
        if g is not None:
            buf.add_(g)  # (fp32 += the gradient of the frame's other uses)
        return buf.to(dt), None


def grad_bank(x: torch.Tensor) -> torch.Tensor:
    """x as a key / value frame of later ltam_attention calls: the same values; the calls' gradients w.r.t. it are summed in one accumulator."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return x
    bank = _GradBank(x.dtype)
    y = _Banked.apply(x, bank)
    y._vmg_bank = bank
    return y


class _LTAM(_Fn):
    @staticmethod
    def forward(ctx, q, loc, rpe, decay_v, cfg, *kv):
        heads, wh, ww, scale, banks = cfg
        t = len(kv) // 2
        keys = [x.contiguous() for x in kv[:t]]
        vals = [x.contiguous() for x in kv[t:]]
        q = q.contiguous()
        loc = loc.contiguous()
        rpe_c = rpe.detach().contiguous()
        out, lse = K.ltam_forward(q, keys, vals, loc, rpe_c, decay_v, heads, wh, ww, scale)
        ctx.cfg = cfg
        ctx.t = t
        # the table's gradient: in the deferred weight-gradient mode every call of the pass adds straight into .grad (no zero-filled temporary per call, no
        # AccumulateGrad add per call -- 12 + 11 tiny launches per train step)
        ctx.direct = DEFERRED.direct(rpe) and rpe.dtype == torch.float32 and rpe.is_contiguous()
        if ctx.direct:
            ctx.rpe_param, ctx.gen = rpe, DEFERRED.note_params(rpe)
        ctx.save_for_backward(q, loc, rpe_c, decay_v, out, lse, *keys, *vals)
        return out

    @staticmethod
    def backward(ctx, dout):
        heads, wh, ww, scale, banks = ctx.cfg
        t = ctx.t
        q, loc, rpe, decay_v, out, lse = ctx.saved_tensors[:6]
        keys = list(ctx.saved_tensors[6:6 + t])
        vals = list(ctx.saved_tensors[6 + t:])
        into = []
        for i, b in enumerate(banks):
            if b is None or not ctx.needs_input_grad[5 + i]:
                into.append(None)
                continue
            if b.buf is None:
                b.buf = K.ACC_POOL.take(q.shape, q.device) if q.dtype != torch.float32 else torch.zeros_like(q, dtype=torch.float32)
            into.append(b.buf)
        dq, dk, dv, drpe = K.ltam_backward(q, keys, vals, loc, rpe, decay_v, out, lse, dout, heads, wh, ww, scale, dk_into=into[:t], dv_into=into[t:],
                                           drpe_into=DEFERRED.grad_of(ctx.rpe_param) if ctx.direct else None)
        if ctx.direct:
            DEFERRED.written(ctx.gen, ctx.rpe_param)
            drpe = None
        # banked frames: their _Banked node hands the sum on; the others: fp32 sums rounded to the tensors' dtype here, once
        gkv = [None if into[i] is not None else (g if g.dtype == q.dtype else g.to(q.dtype)) for i, g in enumerate(dk + dv)]
        return (dq, None, drpe, None, None, *gkv)


def ltam_attention(q, keys, vals, loc, rpe, decay_v, heads: int, wh: int, ww: int, scale: float):
    """LTAM_multi_head.forward_wins without the output projection (models/trajectory.py:683-774).  keys / vals that went through grad_bank
    get their gradients accumulated in place across calls."""
    banks = tuple(getattr(x, "_vmg_bank", None) for x in list(keys) + list(vals))
    return _LTAM.apply(q, loc, rpe, decay_v, (heads, wh, ww, float(scale), banks), *keys, *vals)


# ---- 3-D shifted-window attention (models/swin_3d.py) --------------------------------------------------
class _Win3dAttention(_Fn):
    """rWindowAttention.attention for every window / head / time slice (swin_3d.py:167-252) with window partition, roll, padding,
    mask and bias gather folded into the kernel's addressing (vmg_win3d_attn_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, q, kv, bq, bkv, table, heads, wt, shift):
        q, kv = q.contiguous(), kv.contiguous()
        tab = table.detach().contiguous()
        out, lse = K.win3d_attn_forward(q, kv, bq, bkv, tab, heads, wt, shift)
        ctx.cfg = (heads, wt, shift)
        ctx.save_for_backward(q, kv, bq, bkv, tab, out, lse)
        # the q / kv biases belong to Linears whose deferred weight gradient reports them complete: in mode 'deferred' this node adds its share
        # (padded positions) straight into .grad and is counted as an outstanding contribution (functional._DeferredWgrad.note_extra)
        ctx.direct = bq is not None and bkv is not None and DEFERRED.direct(bq, bkv)
        if ctx.direct:
            ctx.params, ctx.gen = (bq, bkv), DEFERRED.note_extra(bq, bkv)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv, bq, bkv, tab, out, lse = ctx.saved_tensors
        heads, wt, shift = ctx.cfg
        if ctx.direct:
            into = tuple(DEFERRED.grad_of(p) for p in ctx.params)
            dq, dkv, dtable, _, _ = K.win3d_attn_backward(q, kv, bq, bkv, tab, out, lse, dout, heads, wt, shift, into=into)
            DEFERRED.extra_written(ctx.gen, *ctx.params)
            return dq, dkv, None, None, dtable, None, None, None
        dq, dkv, dtable, dbq, dbkv = K.win3d_attn_backward(q, kv, bq, bkv, tab, out, lse, dout, heads, wt, shift)
        return dq, dkv, dbq, dbkv, dtable, None, None, None


def win3d_attention(q: torch.Tensor, kv: torch.Tensor, bq, bkv, table: torch.Tensor, heads: int, wt: int, shift) -> torch.Tensor:
    """q (B,D,H,W,C), kv (B,D,H,W,2C): outputs of the q / kv Linears on the un-partitioned feature map -> (B,D,H,W,C)."""
    return _Win3dAttention.apply(q, kv, bq, bkv, table, int(heads), int(wt), tuple(int(v) for v in shift))
