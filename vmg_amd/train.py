"""Training-step closure for the VMG hot path: loss, optimizer groups, data-parallel gradient exchange.

Thin counterpart of the reference's tools/Trainer.py: it reproduces what sits INSIDE the timed step --
CharbonnierLoss(+EdgeLoss) (utils/loss.py:22-79), AdamW with SPyNet in its own lr=0 group (tools/Trainer.py:66-105),
cosine-restart LR (utils/lr_scheduler.py:5-33) -- and replaces DistributedDataParallel by an explicit bucketed
all-reduce (RCCL over xGMI through torch.distributed backend "nccl"; "gloo" on CPU for tests) that is launched
from gradient hooks while backward is still running.
"""
from __future__ import annotations

import ctypes
import math
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------------
# loss (utils/loss.py)
# ---------------------------------------------------------------------------------------------------------
_GAUSS = None


def _gauss_kernel(device, dtype):
    global _GAUSS
    if _GAUSS is None or _GAUSS.device != device or _GAUSS.dtype != dtype:
        k = torch.tensor([[.05, .25, .4, .25, .05]], device=device, dtype=dtype)
        _GAUSS = (k.t() @ k)[None].repeat(3, 1, 1, 1)
    return _GAUSS


def charbonnier_edge_loss(x: torch.Tensor, y: torch.Tensor, eps: float = 1e-12, aux: bool = True, aux_ratio: float = 0.005):
    """mean sqrt((x-y)^2 + eps) + aux_ratio * mean_t Charbonnier(Laplacian(x_t) - Laplacian(y_t)); x, y (B,T,3,H,W)."""
    loss = torch.mean(torch.sqrt((x - y) ** 2 + eps))
    if not aux:
        return loss
    B, T, C, H, W = x.shape
    kern = _gauss_kernel(x.device, x.dtype)

    def gauss(img):
        return F.conv2d(F.pad(img, (2, 2, 2, 2), mode="replicate"), kern, groups=3)

    def lap(img):
        f = gauss(img)
        z = torch.zeros_like(f)
        z[:, :, ::2, ::2] = f[:, :, ::2, ::2] * 4
        return img - gauss(z)

    # the reference loops over frames and averages the per-frame means; frames are equal-sized so this is one mean.
    # The Laplacian (Gaussian blur, zero-stuffed downsample, blur, subtract; replicate padding) is LINEAR, so
    # lap(x) - lap(y) = lap(x - y): one pyramid instead of two (identical up to fp32 rounding, ~1e-7).
    ld = lap((x - y).reshape(B * T, C, H, W))
    return loss + aux_ratio * torch.mean(torch.sqrt(ld ** 2 + eps))


class _CharbonnierEdgeHip(torch.autograd.Function):
    """The same loss on the HIP kernels of csrc/loss.hip (four HBM-bound gathers instead of ~40 torch / MIOpen launches):
    forward keeps the Laplacian of x - y, backward applies the adjoint pyramid.  x: (B,T,3,H,W) fp32 on the GPU."""

    @staticmethod
    def forward(ctx, x, y, eps, aux_ratio):
        from . import hip
        hip.require_cuda(x, y)
        x, y = x.contiguous(), y.contiguous()
        if x.dtype != torch.float32 or y.dtype != torch.float32 or x.shape != y.shape:
            raise hip.HipError("charbonnier_edge_loss: fp32 tensors of one shape expected")
        H, W = x.shape[-2:]
        planes = x.numel() // (H * W)
        lib = hip.lib()
        nb = int(lib.vmg_charbonnier_edge_blocks(planes, H, W))
        a1 = torch.empty(planes, (H + 1) // 2, (W + 1) // 2, dtype=torch.float32, device=x.device)
        ld = torch.empty_like(x)
        partial = torch.empty(nb, 2, dtype=torch.float32, device=x.device)
        hip.check(lib.vmg_charbonnier_edge_fwd(x.data_ptr(), y.data_ptr(), a1.data_ptr(), ld.data_ptr(), partial.data_ptr(), planes, H, W,
                                               float(eps), hip.stream_ptr()), "vmg_charbonnier_edge_fwd")
        sums = partial.double().sum(0)  # fixed-order reduction of the per-block partials: reproducible
        n = x.numel()
        ctx.save_for_backward(x, y, ld, a1)
        ctx.args = (float(eps), float(aux_ratio), n, planes, H, W)
        return ((sums[0] + aux_ratio * sums[1]) / n).float()

    @staticmethod
    def backward(ctx, dl):
        from . import hip
        x, y, ld, a1 = ctx.saved_tensors
        eps, aux_ratio, n, planes, H, W = ctx.args
        dx = torch.empty_like(x)
        hip.check(hip.lib().vmg_charbonnier_edge_bwd(x.data_ptr(), y.data_ptr(), ld.data_ptr(), a1.data_ptr(), dx.data_ptr(), planes, H, W, eps,
                                                     1.0 / n, aux_ratio / n, hip.stream_ptr()), "vmg_charbonnier_edge_bwd")
        # the incoming gradient is a device scalar (1 when the loss is the root): applied on the device, never read on the host
        return dx.mul_(dl), None, None, None


def charbonnier_edge_loss_hip(x: torch.Tensor, y: torch.Tensor, eps: float = 1e-12, aux_ratio: float = 0.005) -> torch.Tensor:
    """charbonnier_edge_loss(x, y, aux=True) on the GPU kernels (utils/loss.py:22-79)."""
    return _CharbonnierEdgeHip.apply(x, y, eps, aux_ratio)


# ---------------------------------------------------------------------------------------------------------
# LR schedule (utils/lr_scheduler.py:5-33)
# ---------------------------------------------------------------------------------------------------------
def cosine_restart_lr(step: int, base_lr: float, T_period: List[int], restarts: Optional[List[int]] = None,
                      weights: Optional[List[float]] = None, eta_min: float = 1e-7) -> float:
    """Closed form of CosineAnnealingLR_Restart (utils/lr_scheduler.py:5-33) at scheduler step `step`: before the first
    restart the period is T_period[0] from step 0 and the peak is base_lr; from restart i (step restarts[i] > 0) on, the
    period is T_period[i+1], counted from the restart, and the peak is base_lr * weights[i].  The reference's recursion
    multiplies (lr - eta_min) by (1+cos(pi t/T)) / (1+cos(pi (t-1)/T)) per step, which telescopes to this expression (valid
    while t <= T inside a period, which is how every shipped config is set up)."""
    start, T, peak = 0, T_period[0], base_lr
    for i, r in enumerate(restarts or []):
        if r > 0 and step >= r:
            start, T, peak = r, T_period[i + 1], base_lr * (weights[i] if weights else 1.0)
    return eta_min + 0.5 * (peak - eta_min) * (1 + math.cos(math.pi * (step - start) / T))


class LRSchedule:
    """The per-step learning-rate update of the reference's trainer (tools/Trainer.py:244-272) over an optimizer's parameter groups
    (group 0 = SPyNet, group 1 = everything else, further groups e.g. the weight-decay group):

      1. the cosine-restart scheduler's RECURSION (utils/lr_scheduler.py:17-33) -- each step scales the group's CURRENT lr, so whatever the
         steps below wrote into a group is what the next step continues from (the closed form `cosine_restart_lr` holds only while
         nothing else touches the groups);
      2. `reduced_iter` (the reference's recover_flag): from that iteration on group 1 trains at half the scheduled rate -- the scheduled
         value is put back before the scheduler step and halved again after it;
      3. `pre_training`: SPyNet keeps its initial lr (0) while cur_iter <= flow_fix, then follows group 1 at `pre_lr_ratio`;
      4. warm-up: for cur_iter < warmup_iter every group's lr is initial_lr * cur_iter / warmup_iter.

    step(cur_iter) is called once per optimizer step, after it, like Trainer.train_one_sample does."""

    def __init__(self, groups, T_period, restarts=None, weights=None, eta_min: float = 0.0, warmup_iter: int = -1, pre_training: bool = True,
                 flow_fix: Optional[int] = None, pre_lr_ratio: float = 1.0, reduced_iter: Optional[int] = None):
        self.groups = groups
        self.T_period = list(T_period)
        self.restarts = list(restarts) if restarts else [0]
        self.weights = list(weights) if weights else [1]
        if len(self.restarts) != len(self.weights):
            raise ValueError("restarts and their weights do not match.")
        self.eta_min, self.warmup_iter = float(eta_min), int(warmup_iter)
        self.pre_training, self.flow_fix, self.pre_lr_ratio = bool(pre_training), flow_fix, float(pre_lr_ratio)
        self.reduced_iter = reduced_iter
        self.recover = False if reduced_iter is not None else None
        self.past_lr = None
        self.T_max, self.last_restart, self.epoch = self.T_period[0], 0, 0
        for g in groups:
            g.setdefault("initial_lr", g["lr"])
            g["lr"] = g["initial_lr"]  # (the scheduler's construction step)
        if pre_training and (flow_fix is None or len(groups) < 2):
            raise ValueError("pre_training needs flow_fix and the SPyNet group in front of the others")

    def _scheduler_step(self):
        self.epoch += 1
        e = self.epoch
        if e in self.restarts:
            i = self.restarts.index(e)
            self.last_restart, self.T_max = e, self.T_period[i + 1]
            for g in self.groups:
                g["lr"] = g["initial_lr"] * self.weights[i]
            return
        T, k = self.T_max, e - self.last_restart
        if (k - 1 - T) % (2 * T) == 0:
            for g in self.groups:
                g["lr"] = g["lr"] + (g["initial_lr"] - self.eta_min) * (1 - math.cos(math.pi / T)) / 2
            return
        f = (1 + math.cos(math.pi * k / T)) / (1 + math.cos(math.pi * (k - 1) / T))
        for g in self.groups:
            g["lr"] = f * (g["lr"] - self.eta_min) + self.eta_min

    def step(self, cur_iter: int):
        gs = self.groups
        if self.recover:
            gs[1]["lr"] = self.past_lr
        self._scheduler_step()
        if self.recover is not None:
            if cur_iter >= self.reduced_iter:
                self.past_lr, self.recover = gs[1]["lr"], True
                gs[1]["lr"] *= 0.5
            else:
                self.recover = False
        if self.pre_training:
            gs[0]["lr"] = gs[0]["initial_lr"] if cur_iter <= self.flow_fix else gs[1]["lr"] * self.pre_lr_ratio
        if cur_iter < self.warmup_iter:
            for g in gs:
                g["lr"] = g["initial_lr"] / self.warmup_iter * cur_iter
        return [g["lr"] for g in gs]

    def state_dict(self):
        return {k: getattr(self, k) for k in ("T_max", "last_restart", "epoch", "recover", "past_lr")}

    def load_state_dict(self, sd):
        for k, v in sd.items():
            setattr(self, k, v)


# ---------------------------------------------------------------------------------------------------------
# data-parallel gradient exchange
# ---------------------------------------------------------------------------------------------------------
class GradBucketReducer:
    """Bucketed gradient all-reduce overlapped with backward (replaces DDP, tools/Trainer.py:30).

    A gradient-completion hook counts arrivals (autograd's post-accumulate hook, or the deferred batched weight gradient's
    callback); when a bucket is complete an async all_reduce of it is launched on a side stream, so RCCL runs over xGMI
    while the remaining backward kernels run.  finish() makes the compute stream wait for the side stream (no host block
    with RCCL) and averages (ReduceOp.AVG on RCCL; SUM and one divide on gloo, which has no AVG).

    Buckets are contiguous runs of the parameter ORDER, 8 MB by default: small enough that the first exchange starts a few
    milliseconds into backward, large enough that a ring step still moves ~1 MB per xGMI link (7 links x ~153 GB/s per GPU,
    point-to-point: a ring is bound by ONE link, so 8 MB takes ~0.1 ms).  The order starts as reverse registration order and,
    with a FlatAdamW optimizer, is replaced after the first step by the MEASURED completion order of that step: the flat
    parameter / gradient / moment buffers are re-laid out once (FlatAdamW.relayout) so that every bucket is again one
    contiguous slice that is reduced in place -- no pack / unpack copies, and buckets complete in launch order.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 8 << 20, group=None, flat_grad: Optional[torch.Tensor] = None,
                 offsets: Optional[List[int]] = None, optimizer=None, single_rank_collectives: bool = False):
        """flat_grad / offsets (FlatAdamW.g / .offsets, `params` in that order): the gradients already live in one flat
        buffer.  optimizer (a FlatAdamW): the same, plus the one-time re-layout by measured completion order.
        single_rank_collectives: issue every collective even in a process group of ONE rank (by default a lone rank skips them) --
        the whole exchange path (communicator, ReduceOp.AVG, side stream, waits) then runs on a one-GPU box."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (bool(single_rank_collectives) and dist.is_initialized())
        self.bucket_bytes = int(bucket_bytes)
        self.optimizer = optimizer
        if optimizer is not None:
            params, flat_grad, offsets = optimizer.params, optimizer.g, optimizer.offsets
        self.params = [p for p in params if p.requires_grad]
        self.flat_grad = flat_grad
        backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self.op_avg = backend == "nccl"  # RCCL averages in the collective
        self.enabled = True
        self.side = torch.cuda.Stream() if self.params[0].is_cuda else None
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_autograd) for p in self.params]
        self._order_log: List[torch.nn.Parameter] = []
        self._relaid = optimizer is None  # nothing to re-lay out without a flat optimizer
        try:  # conv / linear weights are completed by the deferred batched weight-gradient, not by autograd accumulation
            from . import functional as FH
            FH.DEFERRED.callbacks[:] = [self._on_grad]  # one reducer per process
        except Exception:  # pragma: no cover
            pass
        if flat_grad is not None:
            order = [p for _, p in sorted(zip(offsets, self.params), key=lambda t: -t[0])]  # reverse layout order
        else:
            order = list(reversed(self.params))
        self._build(order, offsets)

    def _build(self, order: List[torch.nn.Parameter], offsets: Optional[List[int]]):
        """Buckets = runs of `order`; with a flat gradient buffer a run must be contiguous in it (any monotone walk is)."""
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, size = [], 0
        for p in order:
            cur.append(p)
            size += p.numel() * 4
            if size >= self.bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        if self.flat_grad is not None:
            off_of = {id(p): o for p, o in zip(self.params, offsets)}
            self.flat = []
            for b in self.buckets:
                lo = min(off_of[id(p)] for p in b)
                hi = max(off_of[id(p)] + (p.numel() + 3) // 4 * 4 for p in b)
                if hi - lo != sum((p.numel() + 3) // 4 * 4 for p in b):
                    raise RuntimeError("GradBucketReducer: a bucket is not one contiguous slice of the flat gradient buffer")
                self.flat.append(self.flat_grad[lo:min(hi, self.flat_grad.numel())])
        else:
            self.flat = [torch.zeros(sum(p.numel() for p in b), dtype=torch.float32, device=b[0].device) for b in self.buckets]
        self.bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                self.bucket_of[p] = bi
        self.reset()

    def reset(self):
        self.pending = [len(b) for b in self.buckets]
        self.works = []
        self._seen = set()

    def _launch(self, bi: int):
        b, flat = self.buckets[bi], self.flat[bi]
        if not self.active:
            return
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream())
            ctx = torch.cuda.stream(self.side)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx:
            if self.flat_grad is None:
                torch._foreach_copy_(list(flat.split([p.numel() for p in b])), [p.grad.reshape(-1) for p in b])
            work = dist.all_reduce(flat, op=dist.ReduceOp.AVG if self.op_avg else dist.ReduceOp.SUM, group=self.group, async_op=True)
            if not self.op_avg and self.side is not None:
                work.wait()  # gloo: host-side wait; then the divide is queued behind it on the side stream
                flat.div_(self.world)
                work = None
        self.works.append((bi, work))

    def _on_autograd(self, p: torch.nn.Parameter):
        """autograd's post-accumulate hook.  It also fires when a backward node returned NO gradient for the parameter (the deferred
        batched weight gradients do): such parameters are complete only when functional.DEFERRED says so (its callback)."""
        try:
            from . import functional as FH
            if id(p) in FH.DEFERRED.managed:
                return
        except Exception:  # pragma: no cover
            pass
        self._on_grad(p)

    def _on_grad(self, p: torch.nn.Parameter):
        if not self.enabled or p not in self.bucket_of or id(p) in self._seen:
            return
        self._seen.add(id(p))
        if not self._relaid:
            self._order_log.append(p)
        bi = self.bucket_of[p]
        self.pending[bi] -= 1
        if self.pending[bi] == 0:
            self._launch(bi)

    def finish(self):
        """Wait for every bucket and leave the averaged gradients in p.grad."""
        if self.active:
            missing = [bi for bi, n in enumerate(self.pending) if n > 0]
            for bi in missing:  # parameters that received no gradient this step still take part (zeros)
                for p in self.buckets[bi]:
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                self._launch(bi)
            for bi, work in self.works:
                if work is not None:
                    work.wait()  # RCCL: the current stream waits for the collective, the host does not block
                b, flat = self.buckets[bi], self.flat[bi]
                if not self.op_avg and self.side is None:
                    flat.div_(self.world)
                if self.flat_grad is None:
                    if self.side is not None:
                        torch.cuda.current_stream().wait_stream(self.side)
                    torch._foreach_copy_([p.grad.reshape(-1) for p in b], list(flat.split([p.numel() for p in b])))
            if self.side is not None:
                torch.cuda.current_stream().wait_stream(self.side)
        self.reset()

    def relayout_by_completion(self) -> bool:
        """After the first full step (call it once the optimizer has consumed the gradients): re-home the flat buffers in the
        measured gradient-completion order of that step and rebuild the buckets on it.  Returns True if it did."""
        if self._relaid or self.optimizer is None or not self._order_log:
            return False
        seen = {id(p) for p in self._order_log}
        order = self._order_log + [p for p in self.params if id(p) not in seen]  # never-completed parameters go last
        if self.active:
            # every rank must cut the SAME buckets: rank 0's measured order is the order (a data-dependent branch -- frames_mirror -- or a
            # parameter without a gradient on one rank would otherwise give equal-sized buckets that hold different parameters, and the
            # in-place all-reduce would average unrelated tensors without any error).  torch DDP broadcasts its rebuilt buckets likewise.
            idx_of = {id(p): i for i, p in enumerate(self.params)}  # (self.params: the optimizer's construction order, equal on all ranks)
            dev = self.flat_grad.device if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
            idx = torch.tensor([idx_of[id(p)] for p in order], dtype=torch.int64, device=dev)
            dist.broadcast(idx, 0, group=self.group)
            order = [self.params[i] for i in idx.tolist()]
            if sorted(idx.tolist()) != list(range(len(self.params))):
                raise RuntimeError("GradBucketReducer: rank 0's completion order is not a permutation of the parameters")
        self._order_log = []
        self._relaid = True
        self.optimizer.relayout(order)
        self.params, self.flat_grad = self.optimizer.params, self.optimizer.g
        self._build(list(self.optimizer.params), self.optimizer.offsets)
        return True


def broadcast_module_state(module: torch.nn.Module, src: int = 0, group=None, single_rank_collectives: bool = False):
    """Parameters AND buffers from rank `src` (what DDP does at wrap time; gamma_h/w, decay_v, spynet.mean/std included)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not single_rank_collectives):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src, group=group)


# ---------------------------------------------------------------------------------------------------------
# optimizer: AdamW over one flat buffer
# ---------------------------------------------------------------------------------------------------------
class FlatAdamW:
    """torch.optim.AdamW's update (tools/Trainer.py:86-105 builds the reference's optimizer) over ONE flat fp32 buffer.

    All parameters are re-homed as views of `self.p` (group-major, registration order inside a group), their gradients
    are persistent views of `self.g`, the moments live in `self.m` / `self.v`.  A step is one HIP kernel per parameter
    group (vmg_adamw_flat: 28 bytes per parameter, HBM-bound) instead of ~26 multi-tensor launches over 560 tensors, and
    the data-parallel exchange all-reduces slices of `self.g` in place (no pack / unpack copies).  The per-step scalars
    travel through a small device tensor, so the step can be captured in a hipGraph.  GPU only (no CPU path)."""

    def __init__(self, groups, lr: float = 2e-4, betas=(0.9, 0.99), eps: float = 1e-8, weight_decay: float = 0.0):
        from . import hip
        self.groups = []
        params = []
        for g in groups:
            ps = [p for p in g["params"] if p.requires_grad]
            self.groups.append({"params": ps, "lr": float(g.get("lr", lr)), "weight_decay": float(g.get("weight_decay", weight_decay)), "initial_lr": float(g.get("lr", lr))})
            params += ps
        if not params:
            raise ValueError("FlatAdamW: no parameters")
        dev = params[0].device
        hip.require_cuda(*params)
        if any(p.dtype != torch.float32 for p in params):
            raise hip.HipError("FlatAdamW keeps fp32 master parameters")
        self.betas, self.eps, self.t = (float(betas[0]), float(betas[1])), float(eps), 0
        # every tensor starts on a 16-byte boundary of the flat buffers (4 floats)
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.n = total
        self.p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.g = torch.zeros_like(self.p)
        self.m = torch.zeros_like(self.p)
        self.v = torch.zeros_like(self.p)
        self.params, self.offsets = params, offs
        self.construction_order = list(params)  # what state_dict() is keyed on: group-major registration order, untouched by relayout()
        with torch.no_grad():
            for p, o in zip(params, offs):
                view = self.p[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.g[o:o + p.numel()].view_as(p)
        # group segments [start, end) in the flat buffer
        k = 0
        for g in self.groups:
            n = len(g["params"])
            g["start"] = offs[k] if n else total
            g["end"] = (offs[k + n] if k + n < len(offs) else total) if n else total
            k += n
        self.hyper = torch.zeros(len(self.groups), 4, dtype=torch.float32, device=dev)
        # pinned staging ring: the upload is asynchronous, so a row is not rewritten until 64 steps later
        self._ring = torch.zeros(64, len(self.groups), 4, dtype=torch.float32).pin_memory()

    @property
    def param_groups(self):
        return self.groups

    @torch.no_grad()
    def relayout(self, order):
        """Re-home parameters, gradients and moments so that, inside every parameter group, tensors follow `order` (a list of
        all parameters, e.g. the measured gradient-completion order of a step).  One-time copy; group segments stay contiguous."""
        rank = {id(p): i for i, p in enumerate(order)}
        new_params = []
        for g in self.groups:
            g["params"] = sorted(g["params"], key=lambda p: rank.get(id(p), len(rank)))
            new_params += g["params"]
        old_off = {id(p): o for p, o in zip(self.params, self.offsets)}
        offs, total = [], 0
        for p in new_params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        assert total == self.n
        np_, ng, nm, nv = (torch.zeros_like(self.p) for _ in range(4))
        for p, o in zip(new_params, offs):
            oo, n = old_off[id(p)], p.numel()
            for dst, src in ((np_, self.p), (ng, self.g), (nm, self.m), (nv, self.v)):
                dst[o:o + n].copy_(src[oo:oo + n])
        self.p, self.g, self.m, self.v = np_, ng, nm, nv
        self.params, self.offsets = new_params, offs
        for p, o in zip(new_params, offs):
            p.data = self.p[o:o + p.numel()].view_as(p)
            p.grad = self.g[o:o + p.numel()].view_as(p)
        k = 0
        for g in self.groups:
            n = len(g["params"])
            g["start"] = offs[k] if n else total
            g["end"] = (offs[k + n] if k + n < len(offs) else total) if n else total
            k += n
        from . import functional as FH
        FH.bump_weight_epoch()  # parameter storage moved: cached weight packs are stale

    def zero_grad(self, set_to_none: bool = False):
        """Gradients are persistent views of the flat buffer: one memset (set_to_none is accepted and ignored)."""
        self.g.zero_()
        views = getattr(self, "_grad_views", None)
        if views is None or len(views) != len(self.params) or self._grad_views_of is not self.g:
            views = self._grad_views = [None] * len(self.params)
            self._grad_views_of = self.g
        for i, p in enumerate(self.params):
            if p.grad is not views[i]:  # (identity test: 560 data_ptr() comparisons cost 0.4 ms at every step boundary, GPU idle behind them)
                o = self.offsets[i]
                views[i] = p.grad = self.g[o:o + p.numel()].view_as(p)

    def _upload(self):
        b1, b2 = self.betas
        host = self._ring[self.t % 64]
        for i, g in enumerate(self.groups):
            host[i, 0] = g["lr"]
            host[i, 1] = g["weight_decay"]
            host[i, 2] = 1.0 - b1 ** self.t
            host[i, 3] = math.sqrt(1.0 - b2 ** self.t)
        self.hyper.copy_(host, non_blocking=True)

    def advance(self):
        """Host side of a step: bump the step count and send this step's scalars (call before replaying a captured graph)."""
        self.t += 1
        self._upload()

    def launch(self):
        from . import hip
        lib = hip.lib()
        for i, g in enumerate(self.groups):
            n = g["end"] - g["start"]
            if n <= 0:
                continue
            o = 4 * g["start"]
            hip.check(lib.vmg_adamw_flat(self.p.data_ptr() + o, self.g.data_ptr() + o, self.m.data_ptr() + o, self.v.data_ptr() + o, n,
                                         self.hyper.data_ptr() + 16 * i, self.betas[0], self.betas[1], self.eps, hip.stream_ptr()), "vmg_adamw_flat")

    @torch.no_grad()
    def step(self):
        self.advance()
        self.launch()

    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """torch.nn.utils.clip_grad_norm_(all parameters, max_norm, norm_type=2) on the flat gradient buffer (tools/Trainer.py:141-143,
        166-167): two launches, fixed summation order.  Returns a DEVICE tensor (total norm, applied coefficient); nothing is read on the host."""
        from . import hip
        lib = hip.lib()
        ws = getattr(self, "_clip_ws", None)
        if ws is None:
            ws = self._clip_ws = torch.empty(int(lib.vmg_grad_clip_ws_bytes()), dtype=torch.uint8, device=self.g.device)
            self._clip_out = torch.zeros(2, dtype=torch.float32, device=self.g.device)
        hip.check(lib.vmg_grad_clip_norm(self.g.data_ptr(), self.n, float(max_norm), ws.data_ptr(), self._clip_out.data_ptr(), hip.stream_ptr()),
                  "vmg_grad_clip_norm")
        return self._clip_out

    def state_dict(self):
        """Layout-independent: the moments are saved PER PARAMETER in construction order (group-major registration order), whatever
        relayout() has done to the flat buffers since -- a checkpoint written after the data-parallel re-layout loads into a fresh
        optimizer (registration-order layout) and the other way round."""
        off_of = {id(p): o for p, o in zip(self.params, self.offsets)}
        m, v = [], []
        for p in self.construction_order:
            o, n = off_of[id(p)], p.numel()
            m.append(self.m[o:o + n].clone())
            v.append(self.v[o:o + n].clone())
        return {"t": self.t, "m": m, "v": v, "numel": [p.numel() for p in self.construction_order],
                "groups": [{k: g[k] for k in ("lr", "weight_decay", "initial_lr")} for g in self.groups]}

    def load_state_dict(self, sd):
        numel = [p.numel() for p in self.construction_order]
        if not isinstance(sd.get("m"), (list, tuple)) or list(sd.get("numel", [])) != numel or len(sd["m"]) != len(numel) or len(sd["v"]) != len(numel):
            raise ValueError("FlatAdamW.load_state_dict: the checkpoint's per-parameter moments do not match this optimizer's parameters "
                             "(count / sizes in construction order)")
        if len(sd["groups"]) != len(self.groups):
            raise ValueError("FlatAdamW.load_state_dict: parameter group count differs")
        off_of = {id(p): o for p, o in zip(self.params, self.offsets)}
        with torch.no_grad():
            for p, m, v in zip(self.construction_order, sd["m"], sd["v"]):
                o, n = off_of[id(p)], p.numel()
                if m.numel() != n or v.numel() != n:
                    raise ValueError("FlatAdamW.load_state_dict: moment size mismatch")
                self.m[o:o + n].copy_(m.reshape(-1))
                self.v[o:o + n].copy_(v.reshape(-1))
        self.t = int(sd["t"])
        for g, s_ in zip(self.groups, sd["groups"]):
            g.update(s_)


# ---------------------------------------------------------------------------------------------------------
# the step
# ---------------------------------------------------------------------------------------------------------
class TrainStep:
    """forward + loss + backward (+ gradient all-reduce) + AdamW, as tools/Trainer.py:125-190 does per sample."""

    def __init__(self, model: torch.nn.Module, lr: float = 2e-4, betas=(0.9, 0.99), weight_decay: float = 0.0, eps_loss: float = 1e-12,
                 aux: bool = True, aux_ratio: float = 0.005, spynet_lr: float = 0.0, distributed: bool = False,
                 bucket_bytes: int = 8 << 20, schedule: Optional[dict] = None, grad_clip: Optional[float] = None,
                 single_rank_collectives: bool = False):
        """schedule: keyword arguments of LRSchedule (T_period, restarts, weights, eta_min, warmup_iter, flow_fix, pre_lr_ratio,
        reduced_iter: the `train:` / `network.flow_fix` keys of the reference's configs) -- the learning rates are then updated after
        every optimizer step like Trainer.update_learning_rate does; None keeps them constant.
        grad_clip: max_norm of clip_grad_norm_ over ALL parameters before the optimizer step (train.if_grad_clip / grad_clip_up).
        single_rank_collectives: see GradBucketReducer."""
        self.model = model
        from . import functional as FH
        FH.set_wgrad_mode("deferred")  # batched weight gradients written straight into the flat gradient buffer
        spy = list(model.spynet.parameters())
        spy_ids = {id(p) for p in spy}
        rest = [p for p in model.parameters() if id(p) not in spy_ids]
        groups = [{"params": spy, "lr": spynet_lr}, {"params": rest}]
        if weight_decay > 0:  # third group: '.mlp_blocks.' parameters (models/vmg.py:408-411)
            wd_ids = {id(p) for p in model.mlp_wd_param}
            groups[1]["params"] = [p for p in rest if id(p) not in wd_ids]
            groups.append({"params": [p for p in rest if id(p) in wd_ids], "weight_decay": weight_decay})
        on_gpu = next(model.parameters()).is_cuda
        if on_gpu:
            self.opt = FlatAdamW(groups, lr=lr, betas=betas, weight_decay=0.0)
            self.reducer = GradBucketReducer([], bucket_bytes, optimizer=self.opt, single_rank_collectives=single_rank_collectives) if distributed else None
        else:  # host-side rehearsals only (the model itself has no CPU path)
            self.opt = torch.optim.AdamW(groups, lr=lr, betas=betas, weight_decay=0.0)
            self.reducer = GradBucketReducer(model.parameters(), bucket_bytes, single_rank_collectives=single_rank_collectives) if distributed else None
        self.schedule = LRSchedule(self.opt.param_groups, **schedule) if schedule is not None else None
        self.grad_clip = float(grad_clip) if grad_clip else None
        self.iter = 0          # optimizer steps taken: the `cur_iter` of the reference's update_learning_rate
        self.grad_norm = None  # device tensor (total norm, coefficient) of the last clipped step
        self.graph = None
        self._static = None
        self.grad_hook = None  # optional callable(TrainStep), run after the gradient exchange and before the optimizer (tests, logging)
        self.loss_args = dict(eps=eps_loss, aux=aux, aux_ratio=aux_ratio)
        if distributed:
            broadcast_module_state(model, single_rank_collectives=single_rank_collectives)

    @staticmethod
    def _flush():
        from . import functional as FH
        FH.flush_deferred_wgrads()

    def capture(self, lrs: torch.Tensor, hrs: torch.Tensor, warmup: int = 2, replayer: bool = False, before_capture=None):
        """Capture one whole training step (forward, loss, backward, deferred weight gradients, AdamW) into a hipGraph.

        Replaying the step's ~2 000 launches from a graph removes the Python / launch overhead.  The weight packs are rebuilt INSIDE the
        graph by the one-launch repack that follows the optimizer (functional.repack_all), so the packs the warm-up steps left behind are
        exactly what the first replay needs; packs that are made on demand (weights modified in place during the forward) are recorded
        where they happen.  Single-GPU only: the bucketed all-reduce stays eager."""
        if self.reducer is not None:
            raise RuntimeError("graph capture is for the single-GPU step; the distributed step runs eagerly")
        from . import functional as FH
        self._static = (lrs.clone(), hrs.clone())
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            # warm up until the set of cached weight packs has settled: a pack whose weight is modified in place between optimizer steps (the
            # MorphFC decay) is recognised in the SECOND step and joins the one-launch repack plans in the third; a plan that is rebuilt inside
            # the capture would upload its entries there (a host-to-device copy: not capturable)
            for i in range(max(warmup, 8)):
                stamp = FH._PACK_STAMP[0]
                self._eager(*self._static)
                if i + 1 >= warmup and FH._PACK_STAMP[0] == stamp and FH._VOL_STATE["stamp"] == stamp and FH._PACK_STATE["stamp"] == stamp:
                    break
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if not isinstance(self.opt, FlatAdamW):
            FH.clear_pack_cache()  # (no repack_all in this path: every pack is recorded where it is first needed)
        if before_capture is not None:
            before_capture()  # (e.g. arm the library's HIP-event sampler: event records issued during the capture become nodes of the graph and are
            #                    re-recorded by every replay, so a replayed step can still be timed kernel by kernel)
        self.graph = torch.cuda.CUDAGraph(keep_graph=replayer)
        with torch.cuda.graph(self.graph):
            self._loss = self._eager(*self._static)
        self._replay = None
        if replayer:
            # experiment (bench.py --replay): read the captured launches out of the graph and re-issue them with plain launches
            # (csrc/replay.hip) instead of hipGraphLaunch.  Measured: 497 vs 492 LR-frames/s eager -- the step is bound by the GPU and its
            # ~2 700 kernel boundaries, not by whoever issues the launches.  NOT for training: torch's replay() advances the Philox offset
            # of the captured DropPath masks, this path does not (the masks of the captured step repeat).
            from . import hip
            n_ops, n_k = ctypes.c_int(0), ctypes.c_int(0)
            h = hip.lib().vmg_replay_build(ctypes.c_void_p(int(self.graph.raw_cuda_graph())), ctypes.byref(n_ops), ctypes.byref(n_k))
            if not h:
                raise hip.HipError("vmg_replay_build: " + hip.lib().vmg_last_error().decode())
            self._replay, self._replay_ops, self._replay_kernels = ctypes.c_void_p(h), n_ops.value, n_k.value
        return self

    def replay(self, lrs: Optional[torch.Tensor] = None, hrs: Optional[torch.Tensor] = None) -> torch.Tensor:
        if lrs is not None:
            self._static[0].copy_(lrs)
            self._static[1].copy_(hrs)
        if isinstance(self.opt, FlatAdamW):
            self.opt.advance()  # uploads THIS step's scalars: the learning rates the previous step's schedule.step() left in the groups
        if getattr(self, "_replay", None) is not None:
            from . import hip
            hip.check(hip.lib().vmg_replay_run(self._replay, 0, self._replay_ops, hip.stream_ptr()), "vmg_replay_run")
        else:
            self.graph.replay()
        self._after_update()  # host-side bookkeeping of the step the graph has just run (the capture pass itself skipped it)
        return self._loss

    def _after_update(self):
        """Trainer.update_learning_rate(step) (tools/Trainer.py:153, 176, 189): the NEXT step's learning rates, then cur_iter + 1.  Host-side
        only; runs after every optimizer step, eager or replayed (FlatAdamW.advance() uploads the groups' rates at the start of the next step)."""
        if self.schedule is not None:
            self.schedule.step(self.iter)
        self.iter += 1

    def state_dict(self):
        """What tools/Trainer.py:355-365 saves as the training state ({epoch, iter, scheduler, optimizer}): optimizer moments / step count /
        group rates, the schedule's recursion state and cur_iter -- a resumed run continues the warm-up / flow_fix / cosine position."""
        return {"iter": self.iter, "opt": self.opt.state_dict(), "schedule": None if self.schedule is None else self.schedule.state_dict()}

    def load_state_dict(self, sd):
        if (sd.get("schedule") is None) != (self.schedule is None):
            raise ValueError("TrainStep.load_state_dict: the checkpoint and this step disagree on having a learning-rate schedule")
        self.opt.load_state_dict(sd["opt"])
        if self.schedule is not None:
            self.schedule.load_state_dict(sd["schedule"])
        self.iter = int(sd["iter"])

    def __call__(self, lrs: torch.Tensor, hrs: torch.Tensor, grad_acc: int = 1, update: bool = True) -> torch.Tensor:
        """One sample (tools/Trainer.py:125-190).  grad_acc / update: gradient accumulation as in the reference's `revise_epoch` branch --
        the loss is divided by grad_acc, gradients add up in the flat buffer over the micro-steps, and only the call with update=True
        steps the optimizer, zeroes the gradients and advances the learning rates.  Unlike the reference (DDP all-reduces in every
        micro-step's backward, tools/Trainer.py:160-190) the gradients are exchanged ONCE, during the updating micro-step's backward,
        when the buffer holds the accumulated sum."""
        if self.graph is not None:
            if grad_acc != 1 or not update:
                raise RuntimeError("the captured step is one whole update; gradient accumulation runs eagerly")
            return self.replay(lrs, hrs)
        return self._eager(lrs, hrs, grad_acc, update)

    def _eager(self, lrs: torch.Tensor, hrs: torch.Tensor, grad_acc: int = 1, update: bool = True) -> torch.Tensor:
        if self.reducer is not None:
            self.reducer.enabled = bool(update)
        out = self.model(lrs)
        if out.is_cuda and self.loss_args["aux"]:
            loss = charbonnier_edge_loss_hip(out.float(), hrs.float(), self.loss_args["eps"], self.loss_args["aux_ratio"])
        else:
            loss = charbonnier_edge_loss(out.float(), hrs.float(), **self.loss_args)
        if grad_acc != 1:
            loss = loss / grad_acc
        loss.backward()
        self._flush()
        if update and self.reducer is not None:
            self.reducer.finish()
        if self.grad_clip is not None:
            # the reference clips after EVERY micro-step's backward (tools/Trainer.py:166-167, 179-180), i.e. the partially accumulated sum is
            # rescaled each time -- reproduced as is.  (Data-parallel: its non-updating micro-steps clip the all-reduced partial sum, ours the
            # local one, because the exchange happens once per update -- the documented deviation of __call__.)
            if isinstance(self.opt, FlatAdamW):
                self.grad_norm = self.opt.clip_grad_norm_(self.grad_clip)
            else:
                self.grad_norm = torch.nn.utils.clip_grad_norm_([p for g in self.opt.param_groups for p in g["params"]], self.grad_clip, norm_type=2)
        if not update:
            return loss.detach()
        if self.grad_hook is not None:
            self.grad_hook(self)
        if isinstance(self.opt, FlatAdamW):
            if self.graph is None and not torch.cuda.is_current_stream_capturing():
                self.opt.advance()  # host scalars of this step (a captured graph gets them from replay())
            self.opt.launch()
            from . import functional as FH
            FH.bump_weight_epoch()  # the kernel rewrote the parameters behind autograd's version counters
            FH.repack_all()         # ... and every weight pack is rebuilt by one launch
        else:
            self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        if not (loss.is_cuda and torch.cuda.is_current_stream_capturing()):
            self._after_update()  # (the capture pass runs no kernels: it is not a step; replay() does the bookkeeping per replayed step)
        if self.reducer is not None:
            self.reducer.relayout_by_completion()  # once, after the first step: buckets follow the measured completion order
        return loss.detach()
