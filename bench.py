#!/usr/bin/env python
"""bench.py -- VMG-REDS-few_levels training step (forward + loss + backward + AdamW) on synthetic REDS-shaped clips.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload = BASELINE.json configs[1]: per-GPU batch 4 x 7 x 3 x 64 x 64, 4x SR, bf16 activations (fp32 accumulate,
fp32 master weights), random-init weights, synthetic data resident in HBM before the timed region.  One "step" is one
pass of the hot path over one batch.  Prints ONE JSON line on rank 0 with the whole-job LR-frames/s, the roofline
object of the dominant kernel (live HIP-event timing of the bf16 conv3x3 144->144 weight-streaming kernel) and the CPU
baseline (the oracle, i.e. this repo's CPU restatement, on one step of the same batch).  --workload train_full / infer /
train_vimeo run the per-GPU shards of BASELINE configs[2] / [3] / [4] (the last with bf16 weights).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU, T, H, W = 4, 7, 64, 64
FWD_GFLOP_PER_FRAME = 292.8  # Conv+Linear, measured on the reference (SURVEY section 6 / BASELINE.md section 2)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_FP8_PEAK_TFLOPS = 5000.0   # dense fp8 (block-scaled MFMA), same guide
# --workload: the default is the configuration BASELINE.json's metric is quoted on (configs[1]); the other two are the per-GPU
# shards of configs[2] (full VMG-REDS, 8 clips over 8 GPUs = 1 clip per GPU) and configs[3] (sliding-window inference)
WORKLOADS = {
    "train": dict(cfg="few", batch=4, frames=7, size=64, ch=144,
                  name="VMG-REDS-few_levels train step, per-GPU batch 4x7x3x64x64 -> 4x SR (BASELINE configs[1])"),
    "train_full": dict(cfg="full", batch=1, frames=7, size=64, ch=112,
                       name="VMG-REDS full config train step, per-GPU batch 1x7x3x64x64 -> 4x SR (BASELINE configs[2]: 8 clips over 8 GPUs)"),
    "train_vimeo": dict(cfg="few", batch=1, frames=7, size=(256, 448), ch=144,
                        name="VMG-few_levels train step on the Vimeo stress shape, per-GPU batch 1x7x3x256x448 -> 4x SR, bf16 weights "
                             "(BASELINE configs[4] without its fp8 weights: not built)"),
    "train_swin": dict(cfg="few", batch=4, frames=7, size=64, ch=144, swin=True,
                       name="VMG-REDS-few_levels with temporal_empty = false (3-D shifted-window attention on stage 1, models/swin_3d.py; no shipped config "
                            "enables it, SURVEY T8), train step, per-GPU batch 4x7x3x64x64 -> 4x SR"),
    "infer": dict(cfg="few", batch=1, frames=100, size=128, ch=144,
                  name="VMG-REDS-few_levels sliding-window inference, 100 x 180x320 -> 720x1280, windows 50/25, tiles 128/20 (BASELINE configs[3])"),
}


def build_model(device, wl=None):
    import vmg_amd
    wl = wl or WORKLOADS["train"]
    # no MIOpen find mode (tools/train.py:111 of the reference asks for cudnn.benchmark): its search took 5 minutes of the driver's
    # bench run for the few ops left on PyTorch-ROCm; VMG_AUTOTUNE=1 turns it back on
    torch.backends.cudnn.benchmark = os.environ.get("VMG_AUTOTUNE") == "1"
    from vmg_amd.data import REDS_FEW_LEVELS, REDS_FULL
    torch.manual_seed(0)
    infer = wl["frames"] > 50
    hw = list(wl["size"]) if isinstance(wl["size"], tuple) else [wl["size"]] * 2
    net = dict(REDS_FULL if wl["cfg"] == "full" else REDS_FEW_LEVELS)
    if wl.get("swin"):
        net["temporal_empty"] = False
    m = vmg_amd.VMG(num_frames=50 if infer else wl["frames"], image_size=hw, is_train=not infer, spynet_pretrained=None,
                    compute_dtype=torch.bfloat16, **net)
    m.spynet = vmg_amd.SPyNet(None)  # the configs' SPyNet checkpoint is a download URL: random init, as stated in "data"
    m = m.to(device)
    return m.eval() if infer else m.train()


def cpu_baseline(workload="train", same=None):
    """The oracle (CPU restatement, fp32, the box's host cores) on a BOUNDED sample of the workload's own job, next to the GPU number.
    same = (state dict, LR clip, HR target) of the GPU leg (CPU tensors): the train workload's step then runs on EXACTLY the GPU leg's initial
    weights and clip, and its output / loss are returned for the `parity` object ("PSNR(GPU output, CPU output)").
    train: one whole step of the bench's batch (4 clips x 7 frames x 64x64: forward + loss + backward + AdamW over every parameter, ~20 s);
    train_full / train_swin: one step on ONE clip of 7 frames; train_vimeo: one step on a 1 x 3 x 64 x 112 clip (1/16 of the frame area, 3 of 7
    frames), scaled by pixels; infer: ONE network call on a 5-frame 128 x 128 tile of the 18 x 50-frame calls a sequence needs, scaled."""
    from oracle import cases as C
    from oracle import recipe as R
    from oracle import vmg_oracle as O
    try:
        naff = len(os.sched_getaffinity(0))
    except AttributeError:
        naff = os.cpu_count() or 1
    threads = max(1, min(16, naff))  # the GPU box gives one GPU a 16-core share; never oversubscribe
    torch.set_num_threads(threads)
    full = workload == "train_full"
    mk = C.cfg_reds_full if full else C.cfg_reds_few
    shapes, _ = C.load_fixture(os.path.join(ROOT, "tests", "golden", "vmg_reds_full.npz" if full else "vmg_reds_few_cfg1.npz"))
    base_cfg = mk(T=5)
    if workload == "train_swin":
        import dataclasses
        base_cfg = dataclasses.replace(base_cfg, temporal_empty=False)
        shapes = None
    chunk_of, window_of = R.vmg_chunk_lookup(base_cfg)
    if shapes is None:  # (no fixture holds the swin variant's state-dict shapes: take them from the product's own module, weights by recipe as everywhere)
        import vmg_amd
        from vmg_amd.data import REDS_FEW_LEVELS
        net = dict(REDS_FEW_LEVELS, temporal_empty=False)
        pm = vmg_amd.VMG(num_frames=7, image_size=[64, 64], is_train=True, spynet_pretrained=None, **net)
        pm.spynet = vmg_amd.SPyNet(None)
        shapes = {k: list(v.shape) for k, v in pm.state_dict().items()}
    sd = R.recipe_state_dict(shapes, 0, chunk_of, window_of)
    if same is not None and workload == "train":
        if sorted(same[0]) != sorted(sd):
            raise SystemExit("bench.py: the product's state dict and the oracle's key list differ")
        sd = {k: same[0][k].detach().clone() for k in sd}
    for k, v in sd.items():
        if v.dtype.is_floating_point and not R.is_buffer(k):
            v.requires_grad_(True)
    leaves = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.AdamW(leaves, lr=2e-4, betas=(0.9, 0.99), weight_decay=0.0)

    def cfg_for(t):
        c = mk(T=t)
        if workload == "train_swin":
            import dataclasses
            c = dataclasses.replace(c, temporal_empty=False)
        return c

    keep = {}

    def one_step(b, t, h, w, seed, train=True, xy=None):
        x = R.synthetic_clip(b, t, h, w, seed) if xy is None else xy[0]
        t0 = time.time()
        if train:
            y = R.synthetic_target(x) if xy is None else xy[1]
            t0 = time.time()
            out = O.vmg_forward(sd, cfg_for(t), x, mutate=False, call_index=1)
            loss = O.charbonnier_edge_loss(out, y)
            loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            keep["out"], keep["loss"] = out.detach(), float(loss)
        else:
            with torch.no_grad():
                O.vmg_forward(sd, cfg_for(t), x, mutate=False, call_index=1)
        return time.time() - t0

    if same is not None and workload == "train":
        # (no separate warm-up pass: it would take an AdamW step on the weights the parity statement is about; the thread pool and the
        #  allocator are warmed by a forward-only call on a tiny clip instead)
        t0 = time.time()
        with torch.no_grad():
            O.vmg_forward(sd, cfg_for(3), R.synthetic_clip(1, 3, 64, 64, 7), mutate=False, call_index=1)
        warm = time.time() - t0
    else:
        warm = one_step(1, 3 if workload != "train_swin" else 4, 64, 64, 7, train=workload != "infer")
    print("[bench] cpu_baseline warm-up pass %.1f s" % warm, file=sys.stderr, flush=True)
    if workload == "train":
        dt = one_step(B_PER_GPU, T, 64, 64, 8, xy=None if same is None else (same[1], same[2]))
        value, sample = B_PER_GPU * T / dt, "one train step of the bench's own job%s: %d clips x %d frames x 64x64, fp32, forward+loss+backward+AdamW (%.1f s)" % (
            " on the GPU leg's own initial weights and clip" if same is not None else "", B_PER_GPU, T, dt)
    elif workload in ("train_full", "train_swin"):
        dt = one_step(1, 7, 64, 64, 8)
        value, sample = 7 / dt, "one train step on ONE clip of 7 frames x 64x64 (the bench's batch is %s), fp32, forward+loss+backward+AdamW (%.1f s)" % ("the same" if full else "4 such clips", dt)
    elif workload == "train_vimeo":
        dt = one_step(1, 3, 64, 112, 8)
        value = 3 * (64 * 112) / (256 * 448) / dt
        sample = "one train step on a 1 x 3 x 64 x 112 clip (1/16 of the 256 x 448 frame area, 3 of 7 frames), %.1f s; value = 256x448-frame equivalents per second (work is linear in pixels)" % dt
    else:
        dt = one_step(1, 5, 128, 128, 8, train=False)
        value = 100.0 / (18 * 10 * dt)
        sample = "ONE forward call on a 5-frame 128 x 128 tile (%.1f s); a 100-frame 180 x 320 sequence needs 18 calls of 50 frames: value = 100 / (180 x that time)" % dt
    print("[bench] cpu_baseline sample %.1f s" % dt, file=sys.stderr, flush=True)
    res = {"value": round(value, 4), "unit": "LR-frames/s", "cores": os.cpu_count() or threads, "threads": threads, "affinity_cores": naff, "kind": "port",
           "sample": sample + "; warm-up pass before it: %.1f s" % warm}
    return (res, keep) if same is not None else res


def psnr_u8(a, b):
    """PSNR after the reference's clamp / x255 / round (tools/Tester.py:249-250, utils/metrics.py:11-26)."""
    import math
    qa, qb = (a.clamp(0, 1) * 255).round().double(), (b.clamp(0, 1) * 255).round().double()
    mse = float(((qa - qb) ** 2).mean())
    return float("inf") if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse))


def recorded_roofline(workload):
    """`roofline` of a run replayed from a hipGraph: HIP events cannot be sampled inside a replayed graph on this ROCm, so the dominant kernel's average launch
    duration is the rocprofv3 --kernel-trace average of the SAME command run eagerly, recorded in profiles/r04_e_graph_roofline.json (a constant from a
    committed profile, stated as such -- not a live measurement)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r04_e_graph_roofline.json")) as f:
            rec = json.load(f)[workload]
    except Exception:
        return None
    flops = 2.0 * rec["channels"] * rec["channels"] * 9 * rec["pixels"]
    ach = flops / (rec["avg_launch_us"] * 1e-6) / 1e12
    return {"bound": "mfma", "kernel": rec["kernel"], "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
            "traffic": None, "avg_launch_us": rec["avg_launch_us"], "launches_per_step": rec["launches_per_step"],
            "measured": "recorded: rocprofv3 --kernel-trace average of the eager run of this command, " + rec["source"] + " (a replayed graph cannot be sampled by events)"}


def run_extra(workload, device, steps, warmup, graph):
    """A short timed run of another BASELINE configuration in the same process, AFTER the headline timing (so the driver's default run times
    them too): train_full = the per-GPU shard of configs[2] replayed from a captured hipGraph, infer = configs[3] (one 100-frame sequence per
    step).  Same discipline as the headline: W untimed warm-up steps, K timed steps between synchronisations, inputs resident in HBM."""
    from vmg_amd import infer
    from vmg_amd.data import synthetic_clip, synthetic_target
    from vmg_amd.train import TrainStep
    wl = WORKLOADS[workload]
    B, Tn, S = wl["batch"], wl["frames"], wl["size"]
    model = build_model(device, wl)
    mode = "eager"
    if workload == "infer":
        lrs = synthetic_clip(1, Tn, 180, 320, seed=7, device=device)
        net = infer.GraphedModel(model) if graph else model
        if graph:
            mode = "hipgraph (one captured network call, replayed per tile)"

        def step():
            with torch.no_grad():
                return infer.to_uint8(infer.test_clips(net, lrs, 50, 25, [128, 128], 20, 4))
    else:
        ts = TrainStep(model, lr=2e-4, betas=(0.9, 0.99), aux=True, aux_ratio=0.005)
        lrs = synthetic_clip(B, Tn, S, S, seed=1234, device=device)
        hrs = synthetic_target(lrs, seed=4321)
        if graph:
            ts.capture(lrs, hrs, warmup=max(1, warmup))
            mode = "hipgraph"

        def step():
            return ts(lrs, hrs)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"workload": wl["name"], "value": round(B * Tn * steps / dt, 3), "unit": "LR-frames/s", "ms_per_step": round(dt / steps * 1e3, 2),
            "steps": steps, "warmup": warmup, "launch": mode, "roofline": recorded_roofline(workload) if graph else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="train", help="train = BASELINE configs[1] (the metric's own "
                    "configuration, the default); train_full / infer = the per-GPU shard of configs[2] / configs[3]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fp8", action="store_true", help="VMG(fp8_chains=True): conv1 / conv2 of the recurrent chains' residual blocks in fp8 (e4m3, block-scaled "
                    "MFMA; SURVEY 8f-4 / BASELINE configs[4]) in the forward pass; the roofline object then prices the fp8 kernel against the 5 PFLOP/s fp8 peak")
    ap.add_argument("--no-prof", action="store_true")
    ap.add_argument("--wgrad3-variant", type=int, default=-1, help="vmg_conv_wgrad3_variant (0: round 2's kernel, 1: conv_wgrad3b_kernel; -1 keeps the default): A/B on one box")
    ap.add_argument("--grouped-dense", type=int, default=-1, help="functional.GROUPED_DENSE (0: grouped convolutions as G launches, 1: one launch on the dense block-diagonal pack; -1 keeps the default): A/B on one box")
    ap.add_argument("--win3d-variant", type=int, default=-1, help="vmg_win3d_variant (0: VALU window-attention kernel, 1: MFMA; -1 keeps the default): A/B on one box")
    ap.add_argument("--spynet-edge-fp32", type=int, default=-1, help="override SPyNet.edge_fp32 (0 / 1; -1 keeps the model's default): A/B of its cost")
    ap.add_argument("--no-extras", action="store_true", help="skip the short runs of the other BASELINE configurations that the default "
                    "(--workload train, one GPU) run appends as `extra_workloads`")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (measured +2 %; the live "
                    "HIP-event roofline needs eager launches, so eager is the default)")
    ap.add_argument("--recompute", action="store_true", help="VMG(recompute_chains=True): the recurrent residual chains keep only their inputs and "
                    "are re-run in the backward (SURVEY 8f-4); prints the peak device memory next to the throughput")
    ap.add_argument("--replay", action="store_true", help="with --graph: re-issue the captured launches with plain hipLaunchKernel calls "
                    "(vmg_replay_run) instead of hipGraphLaunch -- a measurement of the launch path only (+1 %), see TrainStep.capture")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start N fresh ranks (one per GPU, RCCL) as CHILD processes and pass rank 0's
        # JSON line through.  This parent never touches the GPU (no HIP call, no torch.cuda.* besides nothing at all).
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE): refusing to report a different job" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the VMG hot path has no CPU fallback")
    # rehearsal on a one-GPU box: VMG_REHEARSE_ONE_GPU=1 puts every rank on device 0 and exchanges gradients over gloo (the
    # driver's real runs use one GPU per rank and RCCL); it exercises the hooks / buckets / streams of the distributed step
    rehearse = os.environ.get("VMG_REHEARSE_ONE_GPU") == "1"
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from vmg_amd import hip, infer
    from vmg_amd.data import synthetic_clip, synthetic_target
    from vmg_amd.train import TrainStep

    wl = WORKLOADS[args.workload]
    B, Tn, S = wl["batch"], wl["frames"], wl["size"]
    parity_gpu = None
    model = build_model(device, wl)
    model.recompute_chains = bool(args.recompute)
    if args.spynet_edge_fp32 >= 0:
        model.spynet.edge_fp32 = bool(args.spynet_edge_fp32)
    model.fp8_chains = bool(args.fp8)
    if args.workload == "infer":
        lrs = synthetic_clip(1, Tn, 180, 320, seed=7 + rank, device=device)

        net = infer.GraphedModel(model) if args.graph else model  # --graph: every network call replayed from one captured graph (same bits)

        def step(_a, _b):
            with torch.no_grad():
                return infer.to_uint8(infer.test_clips(net, lrs, 50, 25, [128, 128], 20, 4))
        hrs = None
        k1_pixels = 2 * 1 * S * S
    else:
        Hh, Ww = S if isinstance(S, tuple) else (S, S)
        lrs = synthetic_clip(B, Tn, Hh, Ww, seed=1234 + rank, device=device)
        hrs = synthetic_target(lrs, seed=4321 + rank)
        if args.workload == "train" and rank == 0 and world == 1 and not args.no_cpu_baseline:
            # parity of the benchmarked batch itself (outside the timed region): the network's output for the bench's own clip and INITIAL weights,
            # call #1, DropPath off (eval mode: the oracle draws no masks) -- at this batch the recurrent chains take the
            # weight-streaming route (M = 32 768 pixels), the one the roofline object prices.  The oracle computes the same call on the host
            # cores as part of the cpu_baseline step below.
            from vmg_amd.train import charbonnier_edge_loss_hip
            sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
            model.eval()
            with torch.no_grad():
                out0 = model(lrs)
                loss0 = float(charbonnier_edge_loss_hip(out0.float(), hrs.float(), 1e-12, 0.005))
            model.train()
            model.load_state_dict(sd0)  # (the call decayed the MorphFC mixer weights once: put the initial weights back)
            parity_gpu = (out0.float().cpu(), loss0, {k: v.cpu() for k, v in sd0.items()}, lrs.cpu(), hrs.cpu())
            del out0, sd0
        step = TrainStep(model, lr=2e-4, betas=(0.9, 0.99), aux=True, aux_ratio=0.005, distributed=distributed)
        k1_pixels = 2 * B * Hh * Ww  # one frame of every clip for BOTH direction sweeps (run in lockstep)
    k1_flops = 2.0 * wl["ch"] * wl["ch"] * 9 * k1_pixels  # algorithmic FLOPs of one launch of the dominant kernel

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = args.graph and not distributed and args.workload != "infer"
    mode = "eager"
    lib = hip.lib()
    if args.wgrad3_variant >= 0:
        lib.vmg_conv_wgrad3_variant(args.wgrad3_variant)
    if args.win3d_variant >= 0:
        lib.vmg_win3d_variant(args.win3d_variant)
    if args.grouped_dense >= 0:
        from vmg_amd import functional as FH_
        FH_.GROUPED_DENSE = bool(args.grouped_dense)
    if use_graph:
        # (the live HIP-event sampler cannot time kernels inside a replayed graph on this ROCm: event records captured with the step -- plain or
        #  hipEventRecordExternal -- return no timing or fail the capture; a graph run therefore reports no roofline object)
        try:
            step.capture(lrs, hrs, warmup=max(1, args.warmup), replayer=args.replay)
            mode = "captured step, plain launches (vmg_replay_run)" if args.replay else "hipgraph"
        except Exception as e:  # capture is an optimisation, never a requirement
            print("[bench] graph capture failed (%s: %s); running eagerly" % (type(e).__name__, str(e)[:200]), file=sys.stderr, flush=True)
            step.graph = None
            torch.cuda.synchronize()
    if not use_graph and torch.backends.cudnn.benchmark:
        step(lrs, hrs)  # MIOpen picks its kernels on the first encounter of each conv shape: keep that search out of the warmup count
        if rank == 0:
            print("[bench] autotune step done", file=sys.stderr, flush=True)
    for _ in range(args.warmup):
        step(lrs, hrs)
    barrier()
    if rank == 0:
        print("[bench] warmup done", file=sys.stderr, flush=True)
    null_us = 0.0
    if not args.no_prof:
        null_us = float(lib.vmg_prof_null_interval_us(50, hip.stream_ptr()))  # event-pair interval of an empty kernel
        hip.check(lib.vmg_prof_select_pixels(hip.ctx(), k1_pixels), "vmg_prof_select_pixels")
        # every 64th launch of the class is bracketed by an event pair (class 1: the bf16 conv3x3 C -> C; 3: the fp8 one).  An event pair is
        # not free on this runtime -- ~8 us of stream time per sampled launch: at every 16th launch (54 samples per train step) the sampling
        # itself cost 0.43 ms per step (A/B against --no-prof); 13 samples per step still give >= 50 per default run
        hip.check(lib.vmg_prof_begin(hip.ctx(), 3 if args.fp8 else 1, 64, 4096), "vmg_prof_begin")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(lrs, hrs)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    if distributed:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    roofline = None
    if not args.no_prof:
        seen, n, ms = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_double(0.0)
        hip.check(lib.vmg_prof_end(hip.ctx(), ctypes.byref(seen), ctypes.byref(n), ctypes.byref(ms)), "vmg_prof_end")
        if n.value > 0:
            raw_us = ms.value / n.value * 1e3
            # `achieved` is priced on the RAW event-pair interval (conservative): it contains the event/dispatch latency that
            # rocprofv3's kernel timestamps exclude.  null_kernel_interval_us (the same event pair around an empty one-wave
            # kernel) bounds that latency: rocprof's average lies between raw - null and raw.
            avg_s = raw_us * 1e-6
            ach = k1_flops / avg_s / 1e12
            traffic = None  # HBM bytes per launch of this kernel from the PMC passes recorded under profiles/ (not measurable live)
            pmc_file = "r03_g_q8_pmc.json" if args.fp8 else "r04_a_k1_pmc.json"
            if wl["ch"] == 144 and k1_pixels == 32768:
                try:
                    with open(os.path.join(ROOT, "profiles", pmc_file)) as f:
                        traffic = int(json.load(f)["traffic_bytes_per_launch"])
                except Exception:
                    pass
            peak = MFMA_FP8_PEAK_TFLOPS if args.fp8 else MFMA_BF16_PEAK_TFLOPS
            kname = ("conv3x3 %d->%d fp8 (e4m3 operands with E8M0 block scales, v_mfma_scale_f32_16x16x128_f8f6f4, fp32 accumulate) on %d px (convq8_kernel: "
                     "conv1 / conv2 of the recurrent residual blocks, forward; writes bf16 rows and the next convolution's fp8 records)" if args.fp8 else
                     "conv3x3 %d->%d bf16 on %d px (forward + input gradient of the recurrent residual chains, both direction sweeps in one launch: the "
                     "weight-streaming kernel conv_ws_kernel; at <= 16 384 pixels -- one clip per GPU -- the K-split kernel, whose 64-pixel x 64-channel "
                     "workgroups fill the chip)") % (wl["ch"], wl["ch"], k1_pixels)
            roofline = {"bound": "mfma", "kernel": kname,
                        "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                        "traffic": traffic, "traffic_source": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE on tools/%s, profiles/%s (the kernel and shape in isolation)" % ("q8_traffic.py" if args.fp8 else "k1_traffic.py", pmc_file),
                        "avg_launch_us": round(avg_s * 1e6, 2), "event_interval_us": round(raw_us, 2),
                        "null_kernel_interval_us": round(null_us, 2), "launches_per_step": seen.value // max(1, args.steps),
                        "samples": n.value}

    if rank == 0:
        print("[bench] timed region done: %.3f s for %d steps" % (dt, args.steps), file=sys.stderr, flush=True)
    frames = world * B * Tn * args.steps
    value = frames / dt
    train = args.workload != "infer"
    line = {
        "metric": ("LR-frames/s (train: forward+loss+backward+AdamW), VMG-REDS%s 4x SR" % ("-few_levels" if wl["cfg"] == "few" else "")) if train
        else "LR-frames/s (sliding-window inference incl. tile blending and uint8 conversion), VMG-REDS-few_levels 4x SR",
        "value": round(value, 3), "unit": "LR-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16 (fp8 e4m3 operands in the recurrent chains' forward block convolutions)" if args.fp8 else "bf16",
        "data": "synthetic (seeded REDS-shaped clips, random-init weights incl. SPyNet)",
        "config": {"workload": wl["name"], "global_batch": world * B, "frames_per_clip": Tn, "lr_size": (list(S) if isinstance(S, tuple) else [S, S]) if train else [180, 320],
                   "parallelism": f"dp{world}", "per_gpu_value": round(value / world, 3), "launch": mode},
        "roofline": roofline,
    }
    if train:
        line["config"]["loss"] = float(loss)
    elif args.graph:
        line["config"]["launch"] = "hipgraph (vmg_amd.infer.GraphedModel: one captured network call, replayed per tile)"
        line["roofline"] = recorded_roofline("infer") if not args.fp8 else None
    if train and mode == "hipgraph" and not args.fp8 and args.workload in ("train", "train_full"):
        line["roofline"] = recorded_roofline(args.workload)
    line["config"]["recompute_chains"] = bool(args.recompute)
    line["config"]["fp8_chains"] = bool(args.fp8)
    line["config"]["peak_device_memory_GB"] = round(torch.cuda.max_memory_allocated(device) / 1e9, 2)
    if args.workload == "train":
        line["config"]["model_tflops"] = round(3 * FWD_GFLOP_PER_FRAME * value / 1e3, 2)
    if args.workload == "train" and world == 1 and not args.no_extras and not args.graph:
        # the other single-GPU configurations, timed by the same run (only the default workload was driver-timed).
        # After the headline's timed region; an extra that fails is reported as such and never takes the headline down with it.
        extras = []
        del step, model
        from vmg_amd import functional as FH
        for name, st_, wu_, gr_ in (("train_full", 8, 2, True), ("infer", 1, 1, True)):
            try:
                FH.clear_pack_cache()
                torch.cuda.empty_cache()
                extras.append(run_extra(name, device, st_, wu_, gr_))
            except Exception as e:
                extras.append({"workload": WORKLOADS[name]["name"], "error": "%s: %s" % (type(e).__name__, str(e)[:300])})
            print("[bench] extra workload %s done" % name, file=sys.stderr, flush=True)
        line["extra_workloads"] = extras
    if rank == 0:
        line["parity"] = None
        if not args.no_cpu_baseline and world == 1:
            if parity_gpu is not None:
                out_gpu, loss_gpu, sd0, x0, y0 = parity_gpu
                line["cpu_baseline"], kept = cpu_baseline(args.workload, same=(sd0, x0, y0))
                out_cpu = kept["out"]
                line["parity"] = {
                    "what": "network output and Charbonnier+edge loss for the bench's own clip and initial weights, forward call #1, DropPath off: "
                            "HIP bf16 (this GPU, the weight-streaming chain route at M = 32768 px) vs the CPU oracle in fp32 (the cpu_baseline step's own forward)",
                    "psnr_db": round(psnr_u8(out_gpu, out_cpu), 2), "max_abs": round(float((out_gpu - out_cpu).abs().max()), 5),
                    "loss_gpu": loss_gpu, "loss_cpu": kept["loss"],
                    "psnr_gpu_vs_target_db": round(psnr_u8(out_gpu, y0), 3), "psnr_cpu_vs_target_db": round(psnr_u8(out_cpu, y0), 3),
                    "stated_tolerance": "bf16: PSNR(hip, oracle) >= 40 dB, loss within 2 % (DESIGN.md section 2)"}
            else:
                line["cpu_baseline"] = cpu_baseline(args.workload)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
