"""Child process of tests/test_distributed_gpu.py: one data-parallel rank of the tiny VMG on the HIP path.

    python tests/dist_child.py <mode> <outdir>     (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment)

All ranks use GPU 0.  Backend: VMG_DIST_BACKEND = 'gloo' (default; two ranks on a one-GPU box) or 'nccl' (= RCCL; ONE rank -- two
ranks cannot share a device on RCCL -- with every collective still issued: communicator creation, broadcast, ReduceOp.AVG all-reduce on the
side stream, stream waits).  VMG_DIST_CASE: the oracle case whose model runs (vmg_tiny_few; vmg_tiny_swin = with the 3-D window attention).
mode 'reducer': vmg_amd.train.TrainStep (FlatAdamW + GradBucketReducer, deferred batched weight gradients).
mode 'ddp'    : what tools/Trainer.py does -- torch DistributedDataParallel + autocast + GradScaler + clip_grad_norm_ + AdamW.
Writes, per step, the state dict the step started from and the exchanged (averaged) gradients to <outdir>/rank<r>.pt."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode, outdir = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    backend = os.environ.get("VMG_DIST_BACKEND", "gloo")
    if backend == "nccl":
        assert world == 1, "one RCCL rank per device"
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import cases as C
    from oracle import recipe as R
    from tests.util import build_product
    from vmg_amd import functional as FH
    from vmg_amd.train import TrainStep
    name = os.environ.get("VMG_DIST_CASE", "vmg_tiny_few")
    case = C.CASES[name]
    cfg = case["cfg"]
    shapes, _ = C.load_fixture(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))
    sd = C.case_state_dict(case, shapes, seed=rank)  # different weights per rank on purpose: the wrap must broadcast rank 0's
    m = build_product(cfg, torch.float32)
    m.load_state_dict(sd)
    m.train()
    x = R.synthetic_clip(1, cfg.num_frames, 64, 64, 60 + rank).cuda()
    y = R.synthetic_target(x.cpu()).cuda()
    log = []
    if mode == "reducer":
        step = TrainStep(m, lr=1e-4, distributed=True, bucket_bytes=16 << 10, single_rank_collectives=world == 1)  # tiny model: several buckets
        def hook(ts):
            log[-1]["grads"] = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
        step.grad_hook = hook
        for it in range(2):
            log.append({"state": {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}})
            step(x[None][0], y)
        nb = len(step.reducer.buckets)
        assert step.reducer.active and step.reducer.op_avg == (backend == "nccl")
    else:
        ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], find_unused_parameters=False)
        spy = list(m.spynet.parameters())
        ids = {id(p) for p in spy}
        opt = torch.optim.AdamW([{"params": spy, "lr": 0.0}, {"params": [p for p in m.parameters() if id(p) not in ids]}], lr=1e-4,
                                betas=(0.9, 0.99), weight_decay=0.0)
        scaler = torch.amp.GradScaler("cuda")
        from vmg_amd.train import charbonnier_edge_loss_hip
        for it in range(2):
            log.append({"state": {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}})
            opt.zero_grad()
            with torch.autocast("cuda"):
                out = ddp(x, None, True)
                loss = charbonnier_edge_loss_hip(out.float(), y.float())
            scaler.scale(loss).backward()
            scaler.unscale_(opt)
            log[-1]["grads"] = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
            torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=1e9, norm_type=2)
            scaler.step(opt)
            scaler.update()
        nb = 0
    torch.save({"log": log, "buckets": nb, "wgrad_mode": FH.DEFERRED.mode, "backend": dist.get_backend()}, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
