"""Data-parallel gradient exchange (vmg_amd.train.GradBucketReducer) on CPU with gloo, world size 2.

The HIP model itself cannot run on CPU (by design), so the reducer is exercised with a small torch module: the logic
under test -- bucketing in reverse registration order, hook-driven launches during backward, averaging, parameters
without gradients, initial broadcast of parameters and buffers -- is model-independent."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(16, 32)
        self.b = torch.nn.Linear(32, 32)
        self.unused = torch.nn.Linear(4, 4)  # never receives a gradient
        self.c = torch.nn.Linear(32, 8)
        self.register_buffer("gamma", torch.ones(3))

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.a(x)))))


def _worker(rank, world, port, q, flat=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vmg_amd.train import GradBucketReducer, broadcast_module_state
    torch.manual_seed(100 + rank)  # different init per rank on purpose
    net = _Net()
    net.gamma.fill_(float(rank + 5))
    broadcast_module_state(net)
    if flat:
        # the layout train.FlatAdamW gives the gradients: persistent views of one flat buffer, every tensor on a 4-float
        # boundary; buckets are slices of it, all-reduced in place
        params = list(net.parameters())
        offs, total = [], 0
        for p_ in params:
            offs.append(total)
            total += (p_.numel() + 3) // 4 * 4
        gflat = torch.zeros(total)
        for p_, o in zip(params, offs):
            p_.grad = gflat[o:o + p_.numel()].view_as(p_)
        red = GradBucketReducer(params, bucket_bytes=512, flat_grad=gflat, offsets=offs)
    else:
        red = GradBucketReducer(net.parameters(), bucket_bytes=512)  # several small buckets
    outs = []
    for step in range(2):
        torch.manual_seed(7 + 10 * step + rank)
        x = torch.randn(5, 16)
        net(x).square().mean().backward()
        red.finish()
        outs.append({n: (p.grad.numpy().copy() if p.grad is not None else None) for n, p in net.named_parameters()})
        if flat:
            assert all(p_.grad.data_ptr() == gflat.data_ptr() + 4 * o for p_, o in zip(params, offs))  # still the views
            gflat.zero_()
        else:
            net.zero_grad(set_to_none=True)
    # plain numpy through the queue (tensor fd-sharing breaks once the producer exits)
    q.put((rank, {n: p.detach().numpy().copy() for n, p in net.named_parameters()}, net.gamma.numpy().copy(), outs, len(red.buckets)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("flat", [False, True])
def test_bucketed_allreduce_matches_manual_average(flat):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, flat)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    t = torch.from_numpy
    (_, params0, gamma0, outs0, nb), (_, params1, gamma1, outs1, _) = res
    params0, params1 = {n: t(v) for n, v in params0.items()}, {n: t(v) for n, v in params1.items()}
    gamma0, gamma1 = t(gamma0), t(gamma1)
    outs0 = [{n: (t(v) if v is not None else None) for n, v in o.items()} for o in outs0]
    outs1 = [{n: (t(v) if v is not None else None) for n, v in o.items()} for o in outs1]
    assert nb >= 3
    # broadcast: identical parameters and buffers, equal to rank 0's
    for n in params0:
        assert torch.equal(params0[n], params1[n])
    assert torch.equal(gamma0, gamma1) and float(gamma0[0]) == 5.0
    # manual reference: average of the two ranks' local gradients
    net = _Net()
    net.load_state_dict({**params0, "gamma": gamma0})
    for step in range(2):
        want = None
        for rank in range(world):
            torch.manual_seed(7 + 10 * step + rank)
            x = torch.randn(5, 16)
            net.zero_grad()
            net(x).square().mean().backward()
            g = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in net.named_parameters()}
            want = g if want is None else {n: want[n] + g[n] for n in g}
        for n in want:
            ref = want[n] / world
            for outs in (outs0, outs1):
                got = outs[step][n]
                assert got is not None and torch.allclose(got, ref, atol=1e-6), n
